"""bench.py's parts: `flops` (algorithmic work), `workloads` (what a step is), `timing` (the timed region, kernel events, clocks), `cpu` (the
CPU-oracle leg).  bench.py itself is the command line and the JSON line."""
