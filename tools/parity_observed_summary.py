"""profiles/rNN_parity_observed.txt from the records the GPU parity checks append to gpurun_out/parity_observed.jsonl
(tests/parity_checks.py: record_observed):  python tools/parity_observed_summary.py gpurun_out/<tag>/parity_observed.jsonl "<header>" """
import collections
import json
import sys


def main():
    path = sys.argv[1]
    recs = [json.loads(l) for l in open(path) if l.strip()]
    print('# worst observed errors of the GPU parity checks (%d records of %s)%s' % (len(recs), path, (': ' + sys.argv[2]) if len(sys.argv) > 2 else ''))
    print('# The asserted tolerances (tests/parity_checks.py) are <= 10 x these.  kind: metric = worst value over all records of that kind')
    worst = collections.defaultdict(dict)
    for r in recs:
        for k, v in r.items():
            if isinstance(v, float) and (k.endswith('_rel') or k.endswith('_off') or k.endswith('_diff')):
                w = worst[r.get('kind', '?')]
                if k not in w or v > w[k][0]:
                    w[k] = (v, r.get('test', '?'))
    for kind in sorted(worst):
        print('%s: %s' % (kind, ', '.join('%s=%.3e' % (k, v[0]) for k, v in sorted(worst[kind].items()))))
    print('# where the worst values come from')
    for kind in sorted(worst):
        for k, (v, t) in sorted(worst[kind].items()):
            print('%s.%s = %.3e  %s' % (kind, k, v, t))


if __name__ == '__main__':
    main()
