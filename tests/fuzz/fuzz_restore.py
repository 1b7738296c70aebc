"""Offline fuzz (not collected by pytest): gap-heavy random FASTA -> reference builder -> the restore kernels' bodies and the
FASTA formatter (tests/emu) against the reference inspector's output.  usage: fuzz_restore.py <seconds>"""
import os, sys, tempfile, time, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'tools'))
import numpy as np
from oracle import oracle as O
from emu import emu
t_end=time.time()+float(sys.argv[1]); it=0; bad=0
while time.time()<t_end:
    rng=np.random.default_rng(7000+it); it+=1
    d=tempfile.mkdtemp(prefix="fr")
    nseq=int(rng.integers(1,8)); recs=[]
    for i in range(nseq):
        parts=[]
        for _ in range(int(rng.integers(1,5))):
            if rng.random()<0.4: parts.append("N"*int(rng.integers(1,40)))
            parts.append(bytes(np.frombuffer(b"ACGT",dtype=np.uint8)[rng.integers(0,4,int(rng.integers(1,400)))]).decode())
        if rng.random()<0.3: parts.append("N"*int(rng.integers(1,20)))
        s="".join(parts)
        if i>0 and rng.random()<0.15: s="N"*int(rng.integers(1,30))
        w=int(rng.choice([20,60,70]))
        recs.append(">s%d desc %d\n%s\n"%(i,i,"\n".join(s[j:j+w] for j in range(0,len(s),w))))
    open(d+"/genomes.fa","w").write("".join(recs))
    open(d+"/conv.tsv","w").write("".join("s%d\t%d\n"%(i,10+i) for i in range(nseq)))
    open(d+"/nodes.dmp","w").write("1\t|\t1\t|\tno rank\n"+"".join("%d\t|\t1\t|\tspecies\n"%(10+i) for i in range(nseq)))
    open(d+"/names.dmp","w").write("1\t|\troot\t|\t\t|\tscientific name\t|\n")
    extra=[]
    if rng.random()<0.4: extra+=["-o",str(int(rng.choice([1,2,3,6])))]
    if rng.random()<0.4: extra+=["-t",str(int(rng.choice([4,6,8])))]
    try:
        O.ref_build(d,threads=1,extra=tuple(extra))
    except Exception as ex:
        subprocess.run(["rm","-rf",d]); continue
    across=int(rng.choice([60,25,0]))
    want=subprocess.run([os.path.join(O.REF_DIR,"centrifuge-inspect-bin"),"-a",str(across),d+"/idx"],capture_output=True).stdout
    e=emu.Emu(d+"/idx")
    ok=True
    for shift in (int(rng.integers(1,5)), int(rng.integers(5,12))):
        e.inspect_fasta(d+"/o.fa", across, shift)
        if open(d+"/o.fa","rb").read()!=want:
            ok=False; bad+=1; print("MISMATCH",it-1,shift,across,d,flush=True); break
    e.close()
    if ok: subprocess.run(["rm","-rf",d])
print("iterations",it,"bad",bad)
