#!/bin/bash
# usage (on the GPU box): tools/part_state.sh <tag>  -- what kind of part is this lease?  Static identity (unique id, vbios, firmware, power
# cap, clock tables, partition modes), the bench line's rate and launch times, and the shader clock UNDER LOAD from counters:
# GRBM_GUI_ACTIVE (cycles the GPU was busy during a dispatch) / the dispatch's duration, for pass A and pass B.  -> gpurun_out/<tag>/part_state.txt
R="$(cd "$(dirname "$0")/.." && pwd)"; tag=${1:-part}; O=$R/gpurun_out/$tag; mkdir -p $O
{
echo "== identity"; for f in unique_id vbios_version current_compute_partition current_memory_partition power_dpm_force_performance_level; do
  for d in /sys/class/drm/card*/device; do [ -r $d/$f ] && echo "$f: $(cat $d/$f)"; done; done
for d in /sys/class/drm/card*/device; do for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk; do [ -r $d/$f ] && echo "$f: $(tr '\n' ' ' < $d/$f)"; done; done
for h in /sys/class/drm/card*/device/hwmon/hwmon*; do for f in power1_cap power1_cap_max power1_cap_default power1_average power1_input temp1_input temp2_input temp3_input freq1_input freq2_input; do
  [ -r $h/$f ] && echo "hwmon $f: $(cat $h/$f)"; done; done
echo "== rocm-smi"; rocm-smi --showuniqueid --showvbios --showfwinfo --showclocks --showpower --showmaxpower --showtemp --showperflevel --showmemvendor 2>&1 | grep -v "^$" | head -90
echo "== amd-smi"; (amd-smi static --vbios --limit --board 2>&1 || true) | head -60
echo "== bench (3 runs)"
cd $R
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-traffic --frames 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); g=d['gpu_state']; r=d['roofline']
print('RUN vbios', g.get('partition',{}).get('vbios'), 'unique_id', g.get('partition',{}).get('unique_id'), ':', round(d['value']), 'it/s  pass A', round(1e3*r['pass_a']['avg_launch_ms'],2), 'us  pass B', round(1e3*r['avg_launch_ms'],2), 'us  hwmon sclk median', g.get('sclk_mhz',{}).get('median'), 'MHz  power median', g.get('power_w',{}).get('median'), 'W  hbm C', g.get('after_each_region',{}).get('hbm_c'), ' mclk', g.get('after_each_region',{}).get('mclk_mhz'))"; done
echo "== clock under load from counters (GRBM_GUI_ACTIVE / dispatch duration)"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_clk -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --frames 0 >/dev/null 2>&1
python - <<PY
import sqlite3, glob
for db in glob.glob("$O/pmc_clk/r_results.db"):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    try:
        q = """select k.name, p.counter_name, avg(p.counter_value), avg(k.end - k.start), count(*) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id
               where k.name like '%fused_%' group by k.name, p.counter_name"""
        rows = list(c.execute(q))
    except sqlite3.Error as e:
        rows = []
        print("(join failed: %s; tables: %s)" % (e, [t for t in tabs if 'kernel' in t or 'pmc' in t][:12]))
    for name, cn, cv, dur, n in rows:
        k = "pass A" if "potential" in name else "pass B"
        print("%s %-16s avg %.4g per dispatch over %d dispatches, avg duration %.1f us -> %.3f GHz" % (k, cn, cv, n, dur / 1e3, cv / dur))
PY
rm -rf $O/pmc_clk
} > $O/part_state.txt 2>&1
tail -25 $O/part_state.txt
