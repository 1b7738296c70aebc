import csv, sys, collections
d=sys.argv[1]
K=list(csv.DictReader(open(d+"/t_kernel_trace.csv")))
M=list(csv.DictReader(open(d+"/t_memory_copy_trace.csv")))
print(collections.Counter(r["Direction"] for r in M))
s=[r for r in K if "k_search2<2, 4, false>" in r["Kernel_Name"]]
t0=int(s[3]["Start_Timestamp"]); t1=int(s[8]["End_Timestamp"])
blit=[r for r in K if "copyBuffer" in r["Kernel_Name"] and t0<=int(r["Start_Timestamp"])<=t1 and int(r["End_Timestamp"])-int(r["Start_Timestamp"])>300000]
print("big blit copies in the timed region:", len(blit))
plan=[r for r in K if "k_scan_write<2>" in r["Kernel_Name"] and t0<=int(r["Start_Timestamp"])<=t1]
print("k_scan_write<2> ms:", [round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,2) for r in plan])
