"""HBM bytes per launch of the hot C entry points from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: python tools/pmc_traffic.py gpurun_out/<tag>   (expects pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/ below it)

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950 FETCH_SIZE
tallies 128-byte read requests at 64 bytes, so streaming reads are doubled (calibrated on adam_ema_kernel, whose traffic is
known exactly: 4 floats read + 4 written per parameter with the EMA shadow in the parameter buffer, 5 + 5 with a separate one)."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ENTRY = {  # C entry point -> device kernels it launches (substring match on the demangled name)
    'hashgrid_bwd': ['scatter_bin_kernel', 'scatter_accum_kernel'],
    'hashgrid_fwd': ['hashgrid_fwd_bal_kernel', 'hashgrid_fwd_xcd_kernel'],
    'adam_ema_step': ['adam_ema_kernel', 'adam_ema_runs_kernel', 'ngp_step_tail_kernel'],
    'mlp_bwd': ['mlp_bwd_fused_kernel'],
    'mlp_fwd': ['mlp_fwd_fixed_kernel', 'ngp_nets_fwd_kernel'],
}


def per_kernel(path, counter):
    vals = defaultdict(list)
    for f in glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter:
                continue
            vals[r['Kernel_Name']].append((int(r.get('Grid_Size', 0) or 0), float(r['Counter_Value'])))
    return vals


def median_train(launches):
    """median over the launches with the most common grid size (drops the occupancy-refresh launches of the same kernel)"""
    grids = defaultdict(list)
    for g, v in launches:
        grids[g].append(v)
    best = max(grids.values(), key=len)
    best = sorted(best)
    return best[len(best) // 2]


def main(root):
    rd = per_kernel(os.path.join(root, 'pmc_FETCH_SIZE'), 'FETCH_SIZE')
    wr = per_kernel(os.path.join(root, 'pmc_WRITE_SIZE'), 'WRITE_SIZE')
    out, detail = {}, {}
    for entry, kernels in ENTRY.items():
        fr = wb = 0.0
        found = False
        for k in kernels:
            for name, launches in rd.items():
                if k in name:
                    fr += median_train(launches) * 1024.0
                    found = True
            for name, launches in wr.items():
                if k in name:
                    wb += median_train(launches) * 1024.0
        if not found:
            continue
        detail[entry] = {'FETCH_SIZE_bytes_raw': fr, 'WRITE_SIZE_bytes': wb, 'read_correction': 2.0,
                         'hbm_bytes_per_launch': 2.0 * fr + wb}
        out[entry] = 2.0 * fr + wb
    out['_detail'] = detail
    # the whole step: every kernel's per-launch median (training-sized launches), summed - what `roofline.hbm_step` divides by the step time
    # (kernels launched more than once per step with one name - fills - are counted once: a slight under-estimate of small launches)
    total = 0.0
    names = set(rd) | set(wr)
    for name in names:
        total += 2.0 * (median_train(rd[name]) * 1024.0 if name in rd else 0.0) + (median_train(wr[name]) * 1024.0 if name in wr else 0.0)
    out['_step_total_bytes'] = total
    # MFMA utilisation of the net kernels, when the third pass exists (pmc_MFMA: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE):
    # busy SIMD-cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); the busy cycles are 32 per v_mfma_f32_16x16x4_f32 exactly
    mf = os.path.join(root, 'pmc_MFMA')
    if os.path.isdir(mf):
        busy, act = per_kernel(mf, 'SQ_VALU_MFMA_BUSY_CYCLES'), per_kernel(mf, 'GRBM_GUI_ACTIVE')
        mm = {}
        for name in busy:
            if 'mlp_' in name and name in act:
                b, a = median_train(busy[name]), median_train(act[name])
                if a > 0:
                    mm[name.split('(')[0].replace('void arcn::', '')] = b / (a / 8.0 * 1024.0)
        out['_mfma_busy'] = mm
    # which kernels were measured: bench.py compares these with the sources it runs on and reports `traffic_stale` on a mismatch
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'arcnerf_amd', 'csrc')
    out['_source_sha256'] = {f: hashlib.sha256(open(os.path.join(csrc, f), 'rb').read()).hexdigest() for f in ('hashgrid.hip', 'mlp.hip', 'optim.hip', 'adam.hpp', 'common.hpp')}
    out['_note'] = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 8 --warmup 4), median over the '
                    'training launches, KiB*1024, summed over the kernels of an entry point (mlp_* = the two nets of a step are '
                    'different launches of one kernel: value is their median). FETCH_SIZE doubled per MI355X_MICROARCH.md '
                    '(gfx950 tallies 128-B read requests at 64 B); calibration: adam_ema_kernel reads 4 and writes 4 floats per '
                    'parameter (EMA shadow in the parameter buffer; 5 + 5 with a separate shadow).')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1])
