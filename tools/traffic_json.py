"""HBM traffic per launch of the dominant kernel (3x3 implicit-GEMM conv) from the two
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, as the TCC block
cannot hold both).  Units/corrections per MI355X_MICROARCH.md "HBM": the counters are in
KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads ->
x2; WRITE_SIZE is taken as reported (uncalibrated).
Usage: traffic_json.py <fetch_summary.csv> <write_summary.csv> <out.json> [commit] [box] [command]"""
import csv
import json
import sys

# every kernel a 3x3 launch of the bench step can be: direct (TAPS=9), sub-pixel (TAPS=4), Winograd
# (8x16 and 16x16 blocks), three-channel image convs
KERNELS = ('conv_mfma_kernel<9,', 'conv_mfma_kernel<4,', 'conv_h2_kernel', 'conv_h2r_kernel', 'wino_conv_kernel', 'wino16s_conv_kernel',
           'conv_thinin_kernel', 'conv_thinout_kernel')


# kernels that belong to a conv launch without being one: the max-|x| pass in front of an fp16 x 2
# Winograd launch, the finish of a split-K / K-sliced launch (the few 1x1 split-K layers of 4^2 ... 16^2
# pixels share that kernel: a slight over-count).  Their bytes are added, their dispatches are not.
AUX = ('wino_amax_kernel', 'conv_splitk_finish')


def per_launch(path, counter):
    calls, total = 0, 0.0
    for r in csv.DictReader(open(path)):
        if r['Counter'] != counter:
            continue
        if any(k in r['Kernel_Name'] for k in KERNELS):
            calls += int(r['Dispatches'])
            total += float(r['Sum'])
        elif any(k in r['Kernel_Name'] for k in AUX):
            total += float(r['Sum'])
    return calls, (total / calls if calls else 0.0)


def one_run(fetch_csv, write_csv, bench_json):
    nf, f = per_launch(fetch_csv, 'FETCH_SIZE')
    nw, w = per_launch(write_csv, 'WRITE_SIZE')
    b = json.load(open(bench_json))
    return {
        # the configuration the passes ran at, as the bench line of the FETCH pass reports it: bench.py
        # looks a run up by exec_batch_size and refuses a file that has none for its leg
        'exec_batch_size': b['config']['exec_batch_size'], 'lanes': b['config']['lanes'],
        'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --pmc-run --steps %d --warmup %d '
                   '--exec-batch %d' % (b['steps'], b['warmup'], b['config']['exec_batch_size']),
        'fetch_launches': nf, 'write_launches': nw,
        'fetch_size_raw_bytes_per_launch': round(f * 1024),
        'fetch_bytes_per_launch': round(f * 1024 * 2),
        'write_bytes_per_launch': round(w * 1024),
        'hbm_bytes_per_launch': round(f * 1024 * 2 + w * 1024),
        'algo_bytes_per_launch_of_that_run': b['roofline'].get('algo_bytes_per_launch'),
    }


# usage (round 6): traffic_json.py <out.json> <commit> <box> (<fetch_summary.csv> <write_summary.csv> <bench.json>)...
out_path, commit, box = sys.argv[1:4]
runs = [one_run(*sys.argv[i:i + 3]) for i in range(4, len(sys.argv) - 2, 3)]
out = {
    'kernel': '3x3 conv launches: conv_h2_kernel<TAPS=9|4,...> / conv_h2r_kernel / conv_mfma_kernel<TAPS=9|4,...> (direct / sub-pixel) + wino_conv_kernel + '
              'wino16s_conv_kernel + conv_thinin/thinout_kernel; the bytes of wino_amax_kernel (max-|x| pass of '
              'the fp16 x 2 Winograd launches) and conv_splitk_finish* are included, per conv launch',
    'commit': commit, 'box': box,
    'runs': runs,
    'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --pmc-run` (warm-up + '
              'timed steps only: every dispatch belongs to the named execution batch); KiB -> bytes; FETCH_SIZE x2 '
              '(gfx950 wide-read under-count, MI355X_MICROARCH.md "HBM"); WRITE_SIZE as reported (uncalibrated)',
}
json.dump(out, open(out_path, 'w'), indent=1)
print(json.dumps(out))
