"""Print the headline numbers of bench.py JSON lines: python tools/bench_line.py file.json [...]  (or one line on stdin)."""
import json
import sys


def show(name, text):
    d = json.loads(text.strip().splitlines()[-1])
    print(name, d["value"], d["unit"], "ms/step", d["ms_per_step"], "one-in-flight", d.get("frames_per_s_one_in_flight"),
          d["passes_serial_ms"])


if len(sys.argv) > 1:
    for path in sys.argv[1:]:
        show(path, open(path).read())
else:
    show("-", sys.stdin.read())
