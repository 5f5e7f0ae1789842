import sys,json
for ln in sys.stdin:
    try: d=json.loads(ln)
    except Exception: print(ln[:200]); continue
    print(d["config"][:90].ljust(90), d.get("us_per_frame", d.get("ms_per_pair", d.get("ms_per_image"))), d["frac_of_8TBps"], d["parity_vs_oracle"])
