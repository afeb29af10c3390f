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
          "| reduce_l0 %s us frac %s | 4096: %s us" % (d["roofline"]["mean_us"] if d["roofline"] else None,
                                                    d["roofline"]["frac"] if d["roofline"] else None, d["roofline_4096"]["mean_us"]))
    print("   ", " ".join("%s=%.1f" % (n, v["mean_us"]) for n, v in k.items()))
