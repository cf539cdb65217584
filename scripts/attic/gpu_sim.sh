cd $GRAFT_REPO_ROOT
for n in 2 4 8; do
  python bench.py --no-cpu-baseline --sim-world $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('sim-world $n: rank-local step %.1f us for %d playlists -> if all ranks equal: %.0f playlists/s aggregate (no comm); kern %.1f us; plan %s' % (d['ms_per_step']*1e3, d['config']['global_batch'], d['value'], r['avg_launch_ms']*1e3, d['config']['plan']))"
done
bash scripts/gpu_prof.sh sim8 --no-cpu-baseline --sim-world 8 --streams 1 2>&1 | grep "calls=" | head -8
