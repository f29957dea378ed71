import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if ('melspec_kernel' in r['Kernel_Name'] or 'melspec_r16_kernel' in r['Kernel_Name'])]
a,b=idx[3],idx[4]
for r in rows[a:b]:
    n=r['Kernel_Name'].split('(')[0].replace('nafp::','').replace('void ','')
    if 'conv_gemm' in n or 'finish' in n:
        print('%-34s grid %6d x %3d x %2d  wg %4s  %7.1f us  scratch %s' % (n[:34], int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']), r['Workgroup_Size_X'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Scratch_Size']))
