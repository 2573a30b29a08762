# trace build whose find_one_kernel marks are kept per (needle row, workgroup): tools/experiments/r6_trace_mid.py
EXTRA = "-DBLURRILY_TRACE"
EDITS = [
("find_kernels.hip",
"(A).phase_clocks[blockIdx.x * 16u + (i_)] = wall_clock64(); } while (0)",
"(blockIdx.y * gridDim.x + blockIdx.x) < 8192u) (A).phase_clocks[(blockIdx.y * gridDim.x + blockIdx.x) * 16u + (i_)] = wall_clock64(); } while (0)"),
("find_kernels.hip",
"#define ONE_MARK(A, i_) do { if ((A).phase_clocks && threadIdx.x == 0)",
"#define ONE_MARK(A, i_) do { if ((A).phase_clocks && threadIdx.x == 0 &&"),
]
