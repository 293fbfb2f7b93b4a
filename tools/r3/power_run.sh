#!/bin/bash
# GPU box: package power, clocks and temperature sampled once a second while bench.py runs back-to-back launches of
# the headline kernel for ~12 s (and idle before / after).  usage: [POWER_ARGS="--bins 4096 --overlap 2048 --rows 65536" POWER_STEPS=24000] tools/r3/power_run.sh > out.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
smi() { rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|junction" | tr -s ' ' | tr '\n' ';'; echo; }
echo "idle before: $(smi)"
python3 $ROOT/bench.py $POWER_ARGS --steps ${POWER_STEPS:-12000} --warmup 5 --prewarm 5 --no-cpu-baseline --no-strict --no-streaming --no-parity > /tmp/power_bench.json 2>/dev/null &
BP=$!
sleep 4
for i in 1 2 3 4 5 6; do echo "under load $i: $(smi)"; sleep 1; done
wait $BP
python3 -c "import json; d=json.loads(open('/tmp/power_bench.json').read().strip().splitlines()[-1]); print('bench: rows/s %.4g, kernel_ms %.4f, frac %.4f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
sleep 2
echo "idle after: $(smi)"
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
