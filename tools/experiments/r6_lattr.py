# trace build with wall-clock marks per latency-mode TASK (find_kernel<..., RANGED>): tools/experiments/r6_trace_lat.py
# BLURRILY_LT = tasks aimed at per workgroup (default rule when unset); BLURRILY_NOLEARN=1 in the environment at BUILD time: no learning sweep
import os
EXTRA = "-DBLURRILY_TRACE"
EDITS = [
("c_abi.hip",
"  const size_t target_tasks = (n < 56 ? 1 : 2) * wgs;",
"  const size_t target_tasks = (getenv(\"BLURRILY_LT\") ? size_t(atoi(getenv(\"BLURRILY_LT\"))) : (n < 56 ? 1 : 2)) * wgs;"),
("kernels/needle_major.inc",
"    if (slot >= n_work) break;\n",
"    if (slot >= n_work) break;\n#define TASK_MARK(i_) do { if (RANGED && A.phase_clocks && tid == 0 && slot < 8192u) A.phase_clocks[slot * 16u + (i_)] = wall_clock64(); } while (0)\n    TASK_MARK(0);\n"),
("kernels/needle_major.inc",
"    PHASE_NEEDLE(10);\n",
"    PHASE_NEEDLE(10);\n    TASK_MARK(1);\n    if (RANGED && A.phase_clocks && tid == 0 && slot < 8192u) { A.phase_clocks[slot * 16u + 5] = nd.T | (uint64_t(range) << 8) | (uint64_t(qs >= w0 && qs < w1) << 20) | (uint64_t(w0) << 24) | (uint64_t(w1) << 36) | (uint64_t(qs) << 48); }\n"),
("kernels/needle_major.inc",
"        if (pass == 0) {\n          compact_pool",
"        TASK_MARK(2 + pass);\n        if (pass == 0) {\n          compact_pool"),
("kernels/needle_major.inc",
"      if (tid == 0) A.part_count[slot] = nres;\n      __syncthreads();\n",
"      if (tid == 0) A.part_count[slot] = nres;\n      __syncthreads();\n      TASK_MARK(4);\n"),
]
if os.environ.get("BLURRILY_NOLEARN"):
    EDITS.append(("kernels/needle_major.inc",
                  "      const bool learn = qs < A.n_windows && !(qs >= w0 && qs < w1);",
                  "      const bool learn = false;"))
