import sys, json
rows = {}
for line in open(sys.argv[1]):
    line=line.strip()
    if not line.startswith("{"): 
        if line.startswith("OFF="): cur=line
        continue
    d=json.loads(line)
    for k in d["roofline"]["all_gemm_instances"]:
        rows.setdefault(k["kernel"],{}).setdefault(cur,[]).append((k["avg_us"],k["launches"]))
    rows.setdefault("STEP us",{}).setdefault(cur,[]).append((d["ms_per_step"] * 1000,1))
for k,v in rows.items():
    print(f"{k:62s}", "  ".join(f"{c}: "+"/".join(f"{a:.1f}" for a,_ in vals) for c,vals in sorted(v.items())), " x", list(v.values())[0][0][1])
