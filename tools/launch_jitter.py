#!/usr/bin/env python3
"""Does the step time of pipelined steps depend on WHEN the host's launches arrive?  (GPU box.)  Runs the headline bench in
child processes with OPUSGPU_LAUNCH_DELAY_US = 0, 50, 100, 200, 300 (og_debug.hpp: the host sleeps that long before each of the
three launches of a step) for both ways of queuing the steps -- one opusgpu_decode_steps_device call for the window, and one
opusgpu_decode_step_device_modes call per step -- and prints ms per step.  Round 2 placed the kernels with a 40 us spin-wait:
a launch that arrived later than that cost 25 % of the step.  usage: python3 tools/launch_jitter.py [steps]"""
import json, os, subprocess, sys
steps = sys.argv[1] if len(sys.argv) > 1 else "20"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for window in ("on", "off"):
    row = []
    for us in (0, 50, 100, 200, 300):
        env = dict(os.environ, OPUSGPU_LAUNCH_DELAY_US=str(us))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", steps, "--warmup", "2", "--no-cpu-baseline",
                              "--no-other-configs", "--window", window], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        row.append("%3d us: %s" % (us, "%.3f ms" % json.loads(line[-1])["ms_per_step"] if line else "failed"))
    print("window %-3s | %s" % (window, " | ".join(row)), flush=True)
