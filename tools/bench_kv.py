#!/usr/bin/env python3
"""Print the headline fields of bench.py's JSON line (stdin): tag, ms per step, kernel times, value."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(tag, d["config"]["config"], round(d["ms_per_step"], 1), {k: round(v, 1) for k, v in d["kernel_ms"].items()}, f'{d["value"]:.4g}')
