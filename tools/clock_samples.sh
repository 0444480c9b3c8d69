for i in 1 2 3; do python bench.py --no-traffic 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); g=d['gpu_state']; r=d['roofline']
print('SAMPLE', round(d['value']), 'it/s  A', round(1e3*r['pass_a']['avg_launch_ms'],1), 'B', round(1e3*r['avg_launch_ms'],1), 'us  sclk', g['sclk_mhz']['median'], g['sclk_mhz']['min'], g['sclk_mhz']['max'], 'MHz  power', g['power_w']['median'], 'W  per-region sclk', g.get('sclk_mhz_per_region'))"; done
cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -12; cat /sys/class/drm/card*/device/power_dpm_force_performance_level 2>/dev/null | head -2
