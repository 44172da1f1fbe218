"""HBM-side traffic of the roofline kernel from the PMC summaries of tools/profile_round.sh.

    python tools/pmc_traffic.py gpurun_out/prof profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch (rocprofv3 derived counters over TCC_EA0_RDREQ / _WRREQ);
on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes (MI355X_MICROARCH.md, HBM), so reads are doubled.
The figure includes Infinity-Cache hits: it is fabric-side traffic of the L2s, an upper bound of HBM traffic.
"""
import json
import re
import sys

KERNEL = 'k_graph_step<false, true>'        # bench.py's roofline kernel (headline config: no edge dropout, training)
NAME = 'k_graph_step'


def counters(path):
    out = {}
    for line in open(path):
        if KERNEL in line:
            for k, v in re.findall(r'(\w+)=([0-9.e+]+)', line):
                out[k] = float(v)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    c = {}
    for f in ('pmc1.txt', 'pmc2.txt'):
        c.update(counters('%s/%s' % (src, f)))
    fetch, write = c['FETCH_SIZE'] * 1024.0, c['WRITE_SIZE'] * 1024.0
    rec = dict(kernel=NAME, symbol=KERNEL, fetch_bytes_reported=fetch, write_bytes=write,
               traffic_bytes=2.0 * fetch + write,
               l2_hit_rate=c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']),
               note='per launch; FETCH_SIZE doubled (gfx950 correction), + WRITE_SIZE; separate --pmc passes of '
                    'bench.py --no-graph --no-overlap (tools/profile_round.sh)')
    json.dump(rec, open(dst, 'w'), indent=1)
    print(rec)


if __name__ == '__main__':
    main()
