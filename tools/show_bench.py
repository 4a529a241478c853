import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f fps, %.1f ms/step" % (d["value"], d["ms_per_step"]))
print("stage", d.get("stage_ms_per_step"))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic")})
for k in ("cpu_baseline", "cpu_baseline_threaded", "pcie_inclusive", "parity_checked"):
    print(k, d.get(k))
for k in ("configs", "workloads", "streams_sweep"):
    print(k, json.dumps(d.get(k), indent=1)[:3500])
