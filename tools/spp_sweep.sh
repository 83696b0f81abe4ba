for a in "--spp 16" "--spp 32" "--spp 64" "--spp 128" "--rb --spp 16" "--rb --spp 64"; do
timeout -k 10 200 python3 bench.py $a --steps 24 --warmup 4 --no-cpu-baseline --no-rm-leg | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$a', 'value', d['value'], 'ms', d['ms_per_step'], 'rays/launch', r.get('rays_per_launch'), 'kernel_ms', r.get('kernel_ms_per_launch'), {k:v for k,v in r.items() if 'mrays' in k or 'per_ray' in k})" || exit 1
done
