"""Offline fuzz (not collected by pytest; build container only: needs perl and /root/reference): random classification
files (taxIDs in and out of the tree, 1-5 rows per read) through centrifuge-kreport / centrifuge-promote against the
reference's Perl scripts.  usage: fuzz_report_tools.py <seconds>"""
import os, sys, tempfile, time, subprocess, shutil, stat, random
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import common
from oracle import oracle as O
B=os.path.join(ROOT,'centrifuge_amd','bin')+'/'
idx=os.path.join(common.golden("synth_small")[0],"idx")
t=tempfile.mkdtemp()
for s in ("centrifuge-kreport","centrifuge-promote"): shutil.copy("/root/reference/"+s,t)
shim=os.path.join(t,"centrifuge-inspect"); open(shim,"w").write('#!/bin/sh\nexec %s/centrifuge-inspect-bin "$@"\n'%O.REF_DIR); os.chmod(shim,0o755)
taxa=[0,1,2,50,100,101,102]+list(range(1000,1024))+[424242,7]
t_end=time.time()+float(sys.argv[1]); it=0; bad=0
while time.time()<t_end:
    rnd=random.Random(it); it+=1
    lines=["readID\tseqID\ttaxID\tscore\t2ndBestScore\thitLength\tqueryLength\tnumMatches"]
    for r in range(rnd.randint(1,60)):
        k=rnd.choice([1,1,1,2,3,5])
        for j in range(k):
            tx=rnd.choice(taxa); lines.append("r%d\t%s\t%d\t%d\t%d\t%d\t%d\t%d"%(r,rnd.choice(["seq1","genus","unclassified","x|y"]),tx,rnd.choice([0,64,400,2025,7225]),rnd.choice([0,64,400]),rnd.randint(0,100),100,k))
    p=os.path.join(t,"in.tsv"); open(p,"w").write("\n".join(lines)+"\n")
    opts=rnd.choice([[],["--no-lca"],["--show-zeros"],["--min-score","300"],["--min-length","50","--no-lca"]])
    w=subprocess.run(["perl",t+"/centrifuge-kreport","-x",idx]+opts+[p],capture_output=True); g=subprocess.run([B+"centrifuge-kreport","-x",idx]+opts+[p],capture_output=True)
    if w.stdout!=g.stdout or w.returncode!=g.returncode: bad+=1; print("KREPORT DIFF",it-1,opts); shutil.copy(p,"/tmp/bad_kreport_%d.tsv"%(it-1))
    lv=rnd.choice(["species","genus","family","lca","superkingdom","bogus"])
    w=subprocess.run(["perl",t+"/centrifuge-promote",idx,p,lv],capture_output=True); g=subprocess.run([B+"centrifuge-promote",idx,p,lv],capture_output=True)
    if w.stdout!=g.stdout: bad+=1; print("PROMOTE DIFF",it-1,lv); shutil.copy(p,"/tmp/bad_promote_%d.tsv"%(it-1))
print("iterations",it,"bad",bad)
