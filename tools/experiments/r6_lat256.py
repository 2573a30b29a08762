# latency mode up to one needle per TWO workgroups (the rule: four)
EDITS = [("c_abi.hip", "n * 4 > wgs || n_windows <= 2) return 1;", "n * 2 > wgs || n_windows <= 2) return 1;")]
