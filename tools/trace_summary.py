import csv,sys,glob
t=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(t)))
t0=min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    if "mrtm" in r["Kernel_Name"] or "k_pm" in r["Kernel_Name"] or "k_abcd" in r["Kernel_Name"]:
        print(r["Kernel_Name"][:58], 'q',r['Queue_Id'], 'start %.2f end %.2f dur %.2f ms'%((int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6), 'wg',r['Workgroup_Size_X'],'grid',r['Grid_Size_X'],'lds',r['LDS_Block_Size'],'vgpr',r['VGPR_Count'])
