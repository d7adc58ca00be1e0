# Board power and clocks while the scoring launch runs back to back (evidence for DESIGN 5.1 "power, not issue slots"):
# bash scripts/power_sample.sh > gpurun_out/<tag>_power.txt
python bench.py --scale-10m off --steps 4000 --warmup 20 --no-cpu-baseline > /tmp/power_bench.json 2> /tmp/power_bench.err &
pid=$!
sleep 4             # import + gallery build + warm-up take 5-8 s; the timed region is 4000 x 3.3 ms = 13 s
for i in $(seq 1 20); do
  echo "--- sample $i"
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2> /dev/null | grep -E "GPU\[0\]" | grep -E -i "power|sclk|temperature \(Sensor junction" | tr -s "\t" " " | tr "\n" ";"; echo
  sleep 1
done
wait $pid
echo "--- idle"
sleep 2
rocm-smi --showpower --showclocks 2> /dev/null | grep -E "GPU\[0\]" | grep -E -i "power|sclk"
echo "--- power cap"
rocm-smi --showmaxpower 2> /dev/null | grep -E "GPU\[0\]" || true
python - <<'PY'
import json
d = json.loads(open('/tmp/power_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print('--- bench of this run: %.1f q/s, %.4f ms/step, launch %.4f ms = %.1f TF, in-kernel clock %.0f MHz' % (
    d['value'], d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['in_kernel_clock_mhz']))
PY
