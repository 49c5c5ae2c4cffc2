import sys,json
d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["frames_per_s_one_in_flight"], d["passes_serial_ms"])
