"""HBM traffic per launch of the dominant kernel (3x3 implicit-GEMM conv) from the two
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, as the TCC block
cannot hold both).  Units/corrections per MI355X_MICROARCH.md "HBM": the counters are in
KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads ->
x2; WRITE_SIZE is taken as reported (uncalibrated).
Usage: traffic_json.py <fetch_summary.csv> <write_summary.csv> <out.json>"""
import csv
import json
import sys


def per_launch(path, counter):
    calls, total = 0, 0.0
    for r in csv.DictReader(open(path)):
        if r['Counter'] == counter and 'conv_mfma_kernel<9,' in r['Kernel_Name']:
            calls += int(r['Dispatches'])
            total += float(r['Sum'])
    return calls, (total / calls if calls else 0.0)


nf, f = per_launch(sys.argv[1], 'FETCH_SIZE')
nw, w = per_launch(sys.argv[2], 'WRITE_SIZE')
out = {
    'kernel': 'conv_mfma_kernel<TAPS=9,...> (all 3x3 instantiations)',
    'fetch_launches': nf, 'write_launches': nw,
    'fetch_size_raw_bytes_per_launch': round(f * 1024),
    'fetch_bytes_per_launch': round(f * 1024 * 2),
    'write_bytes_per_launch': round(w * 1024),
    'hbm_bytes_per_launch': round(f * 1024 * 2 + w * 1024),
    'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over '
              '`bench.py --steps 2 --warmup 1`; KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read '
              'under-count); WRITE_SIZE uncalibrated',
}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out))
