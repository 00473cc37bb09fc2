import os, subprocess, sys, json
ROOT="/root/repo"
for dbg in (0,1,2,4,3,7,8):
    env=dict(os.environ, NANO_HIP_SKIP="29", SKIP_CHILD="1", NANO_ATTN_DBG=str(dbg))
    r=subprocess.run([sys.executable, ROOT+"/tools/skip_probe.py","40","500"],env=env,capture_output=True,text=True)
    try:
        d=json.loads(r.stdout.strip().splitlines()[-1]); print("attn-only dbg",dbg, {k: round((v-41.2)/28,2) for k,v in d.items()}, flush=True)
    except Exception: print(dbg, r.stdout, r.stderr[-300:])
