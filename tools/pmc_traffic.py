"""HBM-side traffic of the roofline kernel from the PMC summaries of tools/profile_round.sh.

    python tools/pmc_traffic.py gpurun_out/prof profiles/r05_pmc_traffic.json [config]

The record is stamped with the hash of the kernel sources (bench.kernel_source_sha) and the commit (IGMC_COMMIT):
bench.py reports ``roofline.traffic`` from it only while the sources it runs are the ones that were profiled.

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch (rocprofv3 derived counters over TCC_EA0_RDREQ / _WRREQ);
on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes (MI355X_MICROARCH.md, HBM), so reads are doubled.
The figure includes Infinity-Cache hits: it is fabric-side traffic of the L2s, an upper bound of HBM traffic.
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NAME = 'k_graph_step'


def counters(path, KERNEL):
    out = {}
    for line in open(path):
        if KERNEL in line:
            for k, v in re.findall(r'(\w+)=([0-9.e+]+)', line):
                out[k] = float(v)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    config = sys.argv[3] if len(sys.argv) > 3 else 'ml_1m'
    # bench.py's roofline kernel: training instantiation, with edge flags when the config has adj-dropout
    name = NAME
    KERNEL = 'k_graph_step2<false, true, true>' if config == 'ml_1m' else 'k_graph_step2<true, true, true>'
    if config == 'ml_100k':                  # cap 200: the one-launch backward of the dense layers (edge dropout: <true>)
        name, KERNEL = 'k_dl_bwd', 'k_dl_bwd<true, 1, false, false>'      # <FLAGS, NG, DENSE3, GS>
    c = {}
    for f in ('pmc1.txt', 'pmc2.txt'):
        c.update(counters('%s/%s' % (src, f), KERNEL))
    if 'FETCH_SIZE' not in c or 'WRITE_SIZE' not in c:
        # (round 5 shipped a copy of the headline record under the ml_100k name: the symbol had gained a template parameter,
        #  nothing matched, the stale file of the previous configuration stayed in place)
        if os.path.exists(dst):
            os.remove(dst)
        raise SystemExit('pmc_traffic: no counters of %s in %s/pmc1.txt / pmc2.txt' % (KERNEL, src))
    from bench import kernel_source_sha
    fetch, write = c['FETCH_SIZE'] * 1024.0, c['WRITE_SIZE'] * 1024.0
    rec = dict(kernel=name, symbol=KERNEL, config=config, src_sha=kernel_source_sha(),
               commit=os.environ.get('IGMC_COMMIT', 'unknown'), fetch_bytes_reported=fetch, write_bytes=write,
               traffic_bytes=2.0 * fetch + write,
               l2_hit_rate=c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']),
               note='per launch; FETCH_SIZE doubled (gfx950 correction), + WRITE_SIZE; separate --pmc passes of '
                    'bench.py in its default configuration (tools/profile_round.sh; rocprofv3 serialises dispatches while it collects counters)')
    json.dump(rec, open(dst, 'w'), indent=1)
    print(rec)


if __name__ == '__main__':
    main()
