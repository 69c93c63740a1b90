"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'][:70]
            if name.startswith('void at::') or 'elementwise' in name:
                continue
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
            dur[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for name, cs in acc.items():
        print(name, ' dispatches', len(next(iter(cs.values()))), ' avg_us %.1f' % (sum(dur[name]) / len(dur[name])))
        for c, v in sorted(cs.items()):
            print('   %-28s %14.0f' % (c, sum(v) / len(v)))


if __name__ == '__main__':
    main()
