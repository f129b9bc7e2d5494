"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md
'HBM') into profiles/pmc_<config>.json: average HBM-side bytes per launch of the dominant kernel.

    python profiles/collect_pmc.py C2 step_kernel gpurun_out/pmc_fetch gpurun_out/pmc_write

Reads the counter CSVs rocprofv3 writes (…_counter_collection.csv).  Units: FETCH_SIZE / WRITE_SIZE
are KiB-like units of 1024 B per the tool's definition; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide streaming reads (the guide says: double it for 16-B/lane reads; other widths
are uncalibrated).  The kernels here read with dword LDS-DMA and dword loads, so both the raw and the
doubled figure are recorded and `traffic_bytes_per_launch` uses the DOUBLED fetch (upper bound).
"""
import csv
import glob
import json
import os
import sys


def avg_counter(d, kernel_sub, counter):
    tot, n = 0.0, 0
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel_sub in row.get('Kernel_Name', '') and row.get('Counter_Name') == counter:
                    tot += float(row['Counter_Value'])
                    n += 1
    return (tot / n if n else None), n


def main(config, kernel_sub, fetch_dir, write_dir, per_step_kernel=None):
    fetch, nf = avg_counter(fetch_dir, kernel_sub, 'FETCH_SIZE')
    write, nw = avg_counter(write_dir, kernel_sub, 'WRITE_SIZE')
    out = {'config': config, 'kernel': kernel_sub, 'launches_fetch': nf, 'launches_write': nw,
           'FETCH_SIZE_avg_raw_KiB': fetch, 'WRITE_SIZE_avg_KiB': write}
    if fetch is not None and write is not None:
        out['fetch_bytes_raw'] = fetch * 1024
        out['fetch_bytes_doubled'] = 2 * fetch * 1024
        out['write_bytes'] = write * 1024
        out['traffic_bytes_per_launch'] = 2 * fetch * 1024 + write * 1024
        if per_step_kernel:
            # a step of many launches of `kernel_sub` (the layer-wise path's GEMMs): launches per step from a kernel
            # that runs once per step, and the step's traffic through `kernel_sub` = launches x the average
            _, n_steps = avg_counter(fetch_dir, per_step_kernel, 'FETCH_SIZE')
            if n_steps:
                out['per_step_kernel'] = per_step_kernel
                out['launches_per_step'] = nf / n_steps
                out['traffic_bytes_per_launch_avg'] = out['traffic_bytes_per_launch']
                out['traffic_bytes_per_launch'] = out['traffic_bytes_per_launch_avg'] * nf / n_steps
                out['note'] = ('traffic_bytes_per_launch is per UPDATE STEP here (all %s launches of a step), to go '
                               'with a roofline whose flops are per step' % kernel_sub)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pmc_%s.json' % config)
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main(*sys.argv[1:6])
