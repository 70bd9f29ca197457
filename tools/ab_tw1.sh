python - <<'PY'
import json,subprocess,os,sys
for rep in range(2):
    for tw in ("0","26"):
        env=dict(os.environ, ZP_NTT_TW1=tw)
        r=subprocess.run([sys.executable,"bench.py","--steps","20","--warmup","3","--no-pipeline","--no-cpu"],env=env,capture_output=True,text=True)
        try:
            d=json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            print("tw1",tw,"ms_per_step",round(d["ms_per_step"],3),"frac",round(d["roofline"]["frac"],4), json.dumps(d["roofline"].get("per_pass",d["roofline"].get("passes","")))[:400],flush=True)
        except Exception as e:
            print("ERR",tw,r.stdout[-500:],r.stderr[-1500:],flush=True)
PY
