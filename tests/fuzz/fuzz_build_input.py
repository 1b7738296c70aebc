"""Offline fuzz (not collected by pytest): awkward random FASTA files (empty records, all-gap sequences, IUPAC codes, CRLF,
blank lines, several files) -> cf_build_describe against the header / names / sequences of the reference builder + inspector.
Run from tests/ (imports test_build_input).  usage: fuzz_build_input.py <seconds>"""
import os, sys, tempfile, time, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'tools'))
import numpy as np
from oracle import oracle as O
from centrifuge_amd import capi
import test_build_input as T
t_end=time.time()+float(sys.argv[1]); it=0; bad=0; skipped=0
while time.time()<t_end:
    rng=np.random.default_rng(90000+it); it+=1
    d=tempfile.mkdtemp(prefix="fb")
    nfiles=int(rng.integers(1,3)); files=[]
    gid=0
    for fi in range(nfiles):
        recs=[]
        for _ in range(int(rng.integers(1,6))):
            kind=rng.random()
            if kind<0.12: body=""                                   # empty record
            elif kind<0.22: body="N"*int(rng.integers(1,30))         # all gaps
            else:
                parts=[]
                for _ in range(int(rng.integers(1,4))):
                    if rng.random()<0.4: parts.append(str(rng.choice(["N","n","-","R","Y"]))*int(rng.integers(1,15)))
                    s=bytes(np.frombuffer(b"ACGT",dtype=np.uint8)[rng.integers(0,4,int(rng.integers(1,120)))]).decode()
                    if rng.random()<0.2: s=s.lower()
                    parts.append(s)
                if rng.random()<0.3: parts.append("N"*int(rng.integers(1,10)))
                body="".join(parts)
            w=int(rng.choice([10,60,1000]))
            nl=str(rng.choice(["\n","\r\n"]))
            lines=nl.join(body[j:j+w] for j in range(0,len(body),w))
            if rng.random()<0.2: lines=lines.replace(nl, nl+nl, 1)
            recs.append(">g%d some text%s%s%s"%(gid,nl,lines,nl if (lines and rng.random()<0.9) else ""))
            gid+=1
        p=d+"/f%d.fa"%fi
        open(p,"w",newline="").write("".join(recs)); files.append(p)
    open(d+"/conv","w").write("x\t1\n"); open(d+"/nodes","w").write("1\t|\t1\t|\tno rank\n"); open(d+"/names","w").write("1\t|\troot\t|\t\t|\tscientific name\t|\n")
    try:
     r=subprocess.run([os.path.join(O.REF_DIR,"centrifuge-build-bin"),"--conversion-table",d+"/conv","--taxonomy-tree",d+"/nodes","--name-table",d+"/names",",".join(files),d+"/ref"],capture_output=True,text=True,timeout=20)
    except subprocess.TimeoutExpired:
        skipped+=1; print("REF TIMEOUT",it-1,d,flush=True); continue
    try:
        desc=capi.build_describe(files); mine_ok=True
    except Exception as ex:
        mine_ok=False; msg=str(ex)
    if r.returncode!=0 or not os.path.exists(d+"/ref.1.cf"):
        skipped+=1
        if mine_ok and r.returncode!=0 and "first reference sequence" not in r.stderr:
            pass
        subprocess.run(["rm","-rf",d]); continue
    if not mine_ok:
        bad+=1; print("MINE FAILED",it-1,msg,d,flush=True); continue
    n,plen,rst=T.ref_header(d+"/ref.1.cf")
    names=subprocess.run([os.path.join(O.REF_DIR,"centrifuge-inspect-bin"),"-n",d+"/ref"],capture_output=True).stdout
    fasta=subprocess.run([os.path.join(O.REF_DIR,"centrifuge-inspect-bin"),d+"/ref"],capture_output=True).stdout
    ok = desc["len"]==n and np.array_equal(desc["plen"],plen) and np.array_equal(desc["rstarts"],rst) and b"".join(x+b"\n" for x in desc["names"])==names and T.reconstruct(desc)==fasta
    if not ok:
        bad+=1; print("MISMATCH",it-1,d,flush=True)
    else: subprocess.run(["rm","-rf",d])
print("iterations",it,"bad",bad,"skipped(ref failed)",skipped)
