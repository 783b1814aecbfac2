import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], j["value"], j["ms_per_step"], j.get("path_mfma_frac"))
for k in j["roofline"]["kernels"]: print("   ", k["kernel"], k["launches_per_step"], k["avg_launch_us"], k["frac_events"])
print("   covered", j["roofline"].get("covered_time_frac"), j["roofline"].get("plan"))
