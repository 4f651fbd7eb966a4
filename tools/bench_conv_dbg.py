import os, sys, subprocess
for dbg in [0, 1, 2, 4, 3, 5, 6, 7]:
    env = dict(os.environ, SRHIP_CONV_DBG=str(dbg))
    out = subprocess.run([sys.executable, "tools/bench_conv.py", "quick"], env=env, capture_output=True, text=True).stdout
    print("dbg", dbg, flush=True)
    print(out, flush=True)
