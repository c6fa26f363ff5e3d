import json,sys
for l in open(sys.argv[1]):
    if not l.startswith("{"): continue
    j=json.loads(l)
    if "rows" not in j: print(j); continue
    r=j["rows"]
    if not r: print(j); continue
    print(j["setting"][:50].ljust(50), "|", " ".join("%s=%.1f"%(k.replace("n_to_bits_hip","enc").replace("bits_to_n_hip","dec"), v["ms"]) for k,v in r.items() if isinstance(v, dict) and "ms" in v), r.get("pin"), r.get("caller_cpu_at_end"))
