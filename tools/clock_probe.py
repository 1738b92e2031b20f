"""Parse a rocprofv3 counter_collection csv: effective clock = GRBM_GUI_ACTIVE / duration."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(dict)
for r in rows:
    k = (r["Dispatch_Id"], r["Kernel_Name"][:40])
    d[k][r["Counter_Name"]] = float(r["Counter_Value"])
    d[k]["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in d.items():
    if v["dur_us"] < 20:
        continue
    ga = v.get("GRBM_GUI_ACTIVE", 0)
    print(k[1], "dur %.1f us" % v["dur_us"], "GUI_ACTIVE %.0f" % ga, "-> %.2f GHz" % (ga / v["dur_us"] / 1e3),
          {a: int(b) for a, b in v.items() if a not in ("dur_us", "GRBM_GUI_ACTIVE")})
