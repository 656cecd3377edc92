import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = d["roofline"]
g = lambda k: (d.get(k) or {}).get("ms_per_pair")
s = d.get("sample") or {}
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"], "kf", r["keyframe_ms_per_step"], "lat", g("latency"), "hi", g("highres"), {k: v["gpu_resident_ms"] for k, v in (s.get("pairs") or {}).items()})
