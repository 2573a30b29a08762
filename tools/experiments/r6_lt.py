# latency mode's task count from the environment (BLURRILY_LT tasks per workgroup; unset: the default rule)
EDITS = [
("c_abi.hip",
"  const size_t target_tasks = (n < 56 ? 1 : 2) * wgs;",
"  const size_t target_tasks = (getenv(\"BLURRILY_LT\") ? size_t(atoi(getenv(\"BLURRILY_LT\"))) : (n < 56 ? 1 : 2)) * wgs;"),
]
