P=$PWD/nerfstudio_amd/libnsamd_prev.so
mkdir -p gpurun_out/ab_final
{
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
for i in 1 2 3; do
  for arm in prev new; do
    if [ $arm = prev ]; then export NSAMD_LIB=$P; else unset NSAMD_LIB; fi
    echo "== driver window, $arm"
    timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['ms_per_step'], j['value'], r['kernel'], r['avg_launch_ms'], (r.get('runner_up') or {}).get('avg_launch_ms'), j['config'].get('param_checksum',{}).get('params')[:8])"
  done
done
unset NSAMD_LIB
echo "== 100 steps, prev / new"
NSAMD_LIB=$P timeout 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"
timeout 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"
} > gpurun_out/ab_final/summary.txt 2>&1
cat gpurun_out/ab_final/summary.txt
