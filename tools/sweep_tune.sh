#!/bin/bash
# bench.py under a list of SWIFTK_TUNE settings on one box: tools/sweep_tune.sh "1:8" "1:4" ...   (key:value[,key:value])
for t in "$@"; do
  SWIFTK_TUNE="$t" python bench.py --steps 10 --warmup 2 --no-extras 2>/dev/null | tail -1 > /tmp/_sweep.json
  python - "$t" <<'PY'
import json, sys
d = json.load(open("/tmp/_sweep.json"))
print("tune", sys.argv[1], "value", round(d["value"], 1), "w1 ms", round(d["roofline"]["avg_launch_ms"], 3), "attn ms", round(d["attention_roofline"]["avg_launch_ms"], 3))
PY
done
