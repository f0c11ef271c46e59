"""Per-launch durations of selected kernels over the last iteration of a rocprofv3 kernel trace."""
import csv, sys
pat = sys.argv[2].split(',')
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(p in r['Kernel_Name'] for p in pat)]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[3])
for r in rows[-n:]:
    name = r['Kernel_Name'].split('(')[0].split('::')[-1][:24]
    print(f"{name:24s} grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) wg={r['Workgroup_Size_X']} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us")
