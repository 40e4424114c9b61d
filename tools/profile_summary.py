"""Condenses the rocprofv3 output directories written by tools/profile.sh into
kernel_stats_<TR>.csv (per-kernel calls / total / average duration), pmc_<TR>.json (per-launch
averages of every collected counter for the bootstrap kernel, its duration in the counter passes, and
the clock / issue figures derived from them) and pmc_traffic.json (HBM bytes per launch; FETCH_SIZE
doubled on gfx950 as MI355X_MICROARCH.md prescribes for 16 B/lane streams tallied at 64 B)."""
import csv
import glob
import json
import os
import sys

KERNELS = {'NTT': 'k_bootstrap', 'FFT': 'k_bootstrap_fft', 'XFFT': 'k_bootstrap_xfft'}   # XFFT = NTT key, exact-fft engine


def find_all(d, suffix):
    return glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True)


def is_kernel(name, kern):
    return (name.startswith('void ' + kern + '<') or name.startswith(kern + '(') or
            name.startswith('void ' + kern + '(') or name == kern)


def pass_counters(d, kern):
    """{counter: average over launches of the per-launch sum}, average kernel duration (ns) in this pass"""
    vals = {}
    for path in find_all(d, 'counter_collection.csv'):
        per = {}
        for r in csv.DictReader(open(path)):
            if not is_kernel(r.get('Kernel_Name', ''), kern):
                continue
            per.setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
            per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
        for c, dd in per.items():
            vals[c] = sum(dd.values()) / len(dd)
    dur = []
    for path in find_all(d, 'kernel_trace.csv'):
        for r in csv.DictReader(open(path)):
            if is_kernel(r.get('Kernel_Name', ''), kern):
                dur.append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    return vals, (sum(dur) / len(dur) if dur else None)


def main(out):
    traffic = {}
    for tr, kern in KERNELS.items():
        stats = find_all(os.path.join(out, 'stats_' + tr), 'kernel_stats.csv')
        if stats:
            rows = list(csv.DictReader(open(stats[0])))
            with open(os.path.join(out, 'kernel_stats_%s.csv' % tr), 'w') as f:
                w = csv.writer(f)
                w.writerow(['kernel', 'calls', 'total_ns', 'average_ns', 'percent'])
                for r in rows:
                    w.writerow([r.get('Name'), r.get('Calls'), r.get('TotalDurationNs'), r.get('AverageNs'),
                                r.get('Percentage')])
        counters, durations = {}, {}
        for d in sorted(glob.glob(os.path.join(out, 'pmc_%s_g*' % tr))):
            if not os.path.isdir(d):
                continue
            vals, dur = pass_counters(d, kern)
            counters.update(vals)
            for c in vals:
                durations[c] = dur
        if not counters:
            continue
        entry = {'kernel': kern, 'workload': 'gate_nand 4096 bits, bench.py --steps 2 --warmup 1',
                 'counters_per_launch': counters, 'kernel_ns_in_counter_pass': durations,
                 'note': 'rocprofv3 --kernel-trace --pmc, one counter group per pass (tools/profile.sh); values are '
                         'sums over all shader engines / waves, averaged over the launches of the kernel'}
        d = {}
        if 'SQ_WAVE_CYCLES' in counters and counters.get('SQ_WAVES'):
            waves = counters['SQ_WAVES']
            t = durations['SQ_WAVE_CYCLES'] * 1e-9
            rounds = max(1.0, waves / 2048.0)          # 2048 waves are resident at a time (8 per CU)
            # SQ_WAVE_CYCLES counts quad-cycles (MI355X_MICROARCH.md): shader cycles a wave was resident
            d['wave_resident_cycles'] = 4.0 * counters['SQ_WAVE_CYCLES'] / waves
            d['shader_clock_ghz_from_wave_cycles'] = d['wave_resident_cycles'] / (t / rounds) * 1e-9
            if 'SQ_INSTS_VALU' in counters:
                d['valu_instructions_per_wave'] = counters['SQ_INSTS_VALU'] / waves
                d['cycles_per_valu_instruction_per_simd'] = d['wave_resident_cycles'] / 2.0 / d['valu_instructions_per_wave']
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
                if c in counters:
                    d['frac_' + c] = counters[c] / counters['SQ_WAVE_CYCLES']
        if 'GRBM_GUI_ACTIVE' in counters:
            d['grbm_gui_active_per_xcd_ghz'] = counters['GRBM_GUI_ACTIVE'] / 8.0 / durations['GRBM_GUI_ACTIVE']
        if 'SQ_BUSY_CYCLES' in counters:
            d['sq_busy_cycles_per_se_ghz'] = counters['SQ_BUSY_CYCLES'] / 32.0 / durations['SQ_BUSY_CYCLES']
        entry['derived'] = d
        json.dump(entry, open(os.path.join(out, 'pmc_%s.json' % tr), 'w'), indent=1, sort_keys=True)
        if 'FETCH_SIZE' in counters and 'WRITE_SIZE' in counters:
            traffic[tr] = {
                'kernel': kern, 'workload': 'gate_nand 4096 bits',
                'fetch_size_kb': counters['FETCH_SIZE'], 'write_size_kb': counters['WRITE_SIZE'],
                'hbm_bytes_per_launch': (2 * counters['FETCH_SIZE'] + counters['WRITE_SIZE']) * 1024,
                'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --steps 2 '
                        '--warmup 1` (tools/profile.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 '
                        'tallies the 128-B requests of 16 B/lane streams at 64 B)'}
        print(tr, json.dumps(d))
    json.dump(traffic, open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1, sort_keys=True)
    print(json.dumps(traffic))


if __name__ == '__main__':
    main(sys.argv[1])
