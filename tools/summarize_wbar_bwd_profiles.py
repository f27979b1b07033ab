"""tools/summarize_wbar_bwd_profiles.py TAG -- reduce the three rocprofv3 passes of tools/profile_wbar_bwd.py
(gpurun_out/TAG_trace, TAG_fetch, TAG_sq: kernel trace, --pmc FETCH_SIZE, --pmc SQ_*) into the summaries kept under
profiles/rNN/ (wbar_bwd_kernel_trace_summary.csv, wbar_bwd_pmc_FETCH_SIZE.csv, wbar_bwd_pmc_SQ.csv)."""
import collections
import csv
import glob
import os
import sys

tag, out = sys.argv[1], sys.argv[2]
shape = {'4194304': (1, 64, 2048), '131072': (1, 32, 512), '262144': (1, 16, 1024), '1024': (256, 16, 4),
         '2097152': (1, 8, 4096), '16777216': (1, 256, 2048)}


def find(d, pattern):
    return glob.glob(os.path.join("gpurun_out", d, "**", pattern), recursive=True)[0]


def short(name):
    return name.split('(')[0].replace('void ', '')


rows = list(csv.DictReader(open(find(tag + "_trace", "*kernel_trace.csv"))))
d = collections.defaultdict(list)
for r in rows:
    if 'wbar_bwd' in r['Kernel_Name']:
        d[(short(r['Kernel_Name']), r['Grid_Size_X'], r['LDS_Block_Size'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open(os.path.join(out, 'wbar_bwd_kernel_trace_summary.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'grid_threads', 'J', 'S', 'D', 'lds_bytes_per_block', 'calls', 'avg_ns', 'median_ns', 'min_ns',
                'grad_w_bytes', 'GB_per_s_at_avg', 'frac_of_8TBs_at_avg', 'GB_per_s_at_median'])
    for (k, g, l), t in d.items():
        J, S, D = shape[g]
        b = J * S * D * D * 4
        t.sort()
        avg = sum(t) / len(t)
        w.writerow([k, g, J, S, D, l, len(t), round(avg), t[len(t) // 2], t[0], b, round(b / avg, 1), round(b / avg / 8000, 3),
                    round(b / t[len(t) // 2], 1)])


def pmc(dirname):
    f = find(dirname, "*counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'wbar_bwd' in r['Kernel_Name']:
            agg[(short(r['Kernel_Name']), r['Grid_Size'], r['VGPR_Count'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


with open(os.path.join(out, 'wbar_bwd_pmc_FETCH_SIZE.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'grid_threads', 'launches', 'FETCH_SIZE_KiB_raw_avg', 'read_bytes_corrected(x2 gfx950)', 'grad_w_bytes',
                'read_over_algorithmic'])
    for (k, g, _), v in pmc(tag + "_fetch").items():
        J, S, D = shape[g]
        b = J * S * D * D * 4
        a = sum(v['FETCH_SIZE']) / len(v['FETCH_SIZE'])
        w.writerow([k, g, len(v['FETCH_SIZE']), round(a, 1), round(2 * a * 1024), b, round(2 * a * 1024 / b, 4)])
names = ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_LDS',
         'GRBM_GUI_ACTIVE']
with open(os.path.join(out, 'wbar_bwd_pmc_SQ.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'grid_threads', 'arch_vgpr_granules_reported'] + names +
               ['VALU_insts_per_wave', 'LDS_insts_per_wave', 'wave_quad_cycles_per_wave', 'wait_frac_of_wave_cycles'])
    for (k, g, vg), v in pmc(tag + "_sq").items():
        m = {n: sum(v[n]) / len(v[n]) for n in names if n in v}
        w.writerow([k, g, vg] + [round(m.get(n, 0)) for n in names] +
                   [round(m['SQ_INSTS_VALU'] / m['SQ_WAVES'], 1), round(m['SQ_INSTS_LDS'] / m['SQ_WAVES'], 1),
                    round(m['SQ_WAVE_CYCLES'] / m['SQ_WAVES'], 1), round(m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], 3)])
for name in ('wbar_bwd_kernel_trace_summary.csv', 'wbar_bwd_pmc_FETCH_SIZE.csv', 'wbar_bwd_pmc_SQ.csv'):
    print(open(os.path.join(out, name)).read())
