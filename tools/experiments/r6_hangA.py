EDITS = [("kernels/needle_major.inc",
"    if (!A.work_list && nd.T > (SHORT ? 64u : 127u)) { // longer needles: the mid / wide-counter launches",
"    if (!A.work_list && (nd.T > (SHORT ? 64u : 127u) || RANGED)) { // EXPERIMENT: every task of a ranged launch skips")]
