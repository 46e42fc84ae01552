"""Condensed view of a bench.py JSON line (file argument)."""
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline %.2f GiB/s, %.1f ms/step, stages %s, chunks checked %s, frac %s" % (j["value"], j["ms_per_step"], j["roofline"]["stage_ms_per_step"],
      j["config"]["chunks_checked_against_reference_hashes"], j["roofline"]["frac"]))
for k, v in j.get("extra", {}).items():
    print("%-24s %8.2f GiB/s  %s ms  ratio %s  %s  %s" % (k, v.get("value"), v.get("ms_per_step", v.get("ms")), v.get("compression_ratio"),
          (v.get("roofline") or {}).get("stage_ms_per_step"), {a: b for a, b in v.items() if a.startswith("first_256") or a in ("bytes_equal", "of_resident_rate", "decoded_in_pieces")}))
cb = j.get("cpu_baseline")
if cb:
    print("cpu_baseline", cb["value"], cb["unit"], cb["cores"], cb["kind"])
