import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for name in ('sample_merge', 'merge_tail', 'gemm_panel_kernel<4, 2', 'xattn_kernel'):
    d = [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if name in r['Kernel_Name']]
    print(name, len(d), 'min', round(min(d)), 'med', round(sorted(d)[len(d)//2]), 'max', round(max(d)), [round(x) for x in d[:60]])
