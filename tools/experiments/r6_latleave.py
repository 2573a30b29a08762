# latency mode's ranges leave dense slices out of a step's count like the throughput sweep does (measured again on the round's final tasks)
EDITS = [
("c_abi.hip", "      a.nm_cmin = 0;                                        // (ranges leave nothing out of a step's count: measured, slower)",
              "      a.nm_cmin = ix.n_bitmaps != 0 && limit <= 64 ? m->nm_cmin : 0u;"),
("c_abi.hip", "      a.pass_base = 0; a.keep = limit; a.pool_cap = find_pool_cap(limit);\n      if (!(a.queue = next_queue())) { errno = EIO; return -1; }\n      // (every task writes its part_count",
              "      a.pass_base = 0; a.keep = limit; a.pool_cap = find_pool_cap(limit);\n      a.nm_cmin = ix.n_bitmaps != 0 && limit <= 64 ? m->nm_cmin : 0u;\n      if (!(a.queue = next_queue())) { errno = EIO; return -1; }\n      // (every task writes its part_count"),
]
