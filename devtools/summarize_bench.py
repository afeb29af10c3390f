"""Print the headline fields of bench.py JSON lines (files given on the command line)."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as ex:  # noqa: BLE001
        print(path, "unreadable:", ex)
        continue
    k = d.get("kernels", {})
    print(path, d["value"], "MP/s", d["ms_per_step"], "ms |", d["config"].get("dispatch"),
          "| metric kernel 4096^2 from HBM %s us frac %s (copy ceiling %s us) | warm %s us | single image %s ms" % (
              d["roofline"]["mean_us"] if d.get("roofline") else None, d["roofline"]["frac"] if d.get("roofline") else None,
              (d["roofline"].get("copy_ceiling") or {}).get("mean_us") if d.get("roofline") else None,
              (d.get("roofline_4096_warm") or {}).get("mean_us"), (d.get("single_image") or {}).get("ms_per_image")))
    print("   ", " ".join("%s=%.1f" % (n, v["mean_us"]) for n, v in k.items()))
