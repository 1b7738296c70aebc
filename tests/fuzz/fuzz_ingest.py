"""Offline fuzz (not collected by pytest): random awkward FASTA / FASTQ read files (empty records, odd names, CRLF, IUPAC and
non-letter characters, trims) through `centrifuge-class --dump-reads` against the read names and lengths the reference binary
reports (run on the CPU).  Differences that remain are malformed FASTQ records (no base letter but a non-empty quality line),
which the reference misparses silently and this front end rejects.  usage: fuzz_ingest.py <seconds>"""
import os, subprocess, sys, tempfile, time, random
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,ROOT)
import common
from oracle import oracle as O
CLI=os.path.join(ROOT,'centrifuge_amd','bin','centrifuge-class')
d,_=common.golden("example")
def ref(args):
    with tempfile.TemporaryDirectory() as t:
        try: r=subprocess.run([os.path.join(O.REF_DIR,"centrifuge-class"),"-x",os.path.join(d,"idx"),"--report-file",os.path.join(t,"r.tsv"),"-S",os.path.join(t,"o.tsv")]+args,capture_output=True,text=True,timeout=20)
        except subprocess.TimeoutExpired: return "TIMEOUT"
        if r.returncode!=0: return "ERR"
        rows=[ln.split("\t") for ln in open(os.path.join(t,"o.tsv")).read().splitlines()[1:]]
    out=[];i=0
    while i<len(rows):
        out.append((rows[i][0],int(rows[i][6]))); i+=max(1,int(rows[i][7]))
    return out
def norm(n):
    n=n.split()[0] if n.split() else n
    if len(n)>=2 and n[-2]=='/' and n[-1] in '123': n=n[:-2]
    return n
def mine(args,env=None):
    r=subprocess.run([CLI,"--dump-reads"]+args,capture_output=True,env=dict(os.environ,CF_DEBUG_KNOBS="1",**(env or {})))   # the CF_* knobs are read only behind this gate (cf_knobs.hpp)
    if r.returncode!=0: return "ERR"
    res=[]
    for ln in r.stdout.split(b"\n")[:-1]:
        f=ln.rsplit(b"\t",3); res.append((norm(f[0].decode("latin1")), len(f[1])))
    return res
ALPH="ACGTACGTACGTNnacgtRY.-"
def seq(rnd,n): return "".join(rnd.choice(ALPH) for _ in range(n))
t_end=time.time()+float(sys.argv[1]); it=0; bad=0
tmp=tempfile.mkdtemp()
while time.time()<t_end:
    rnd=random.Random(5000+it); it+=1
    fq=rnd.random()<0.5; nl=rnd.choice(["\n","\n","\r\n"]); recs=[]
    for i in range(rnd.randint(1,12)):
        name=rnd.choice(["r%d"%i,"r%d desc"%i,"","r%d/1"%i,"x y z","a\tb"])
        L=rnd.choice([0,1,5,30,80,rnd.randint(0,120)])
        s=seq(rnd,L)
        if fq:
            q="".join(chr(rnd.randint(33,73)) for _ in range(len(s)+rnd.choice([0,0,0,0,1])))
            if rnd.random()<0.1: q="@"+q[1:] if q else q
            w=rnd.choice([1000,1000,20,7,1])                       # the sequence over several lines (the reference reads up to the '+')
            body=nl.join(s[j:j+w] for j in range(0,len(s),w)) if w<1000 else s
            recs.append("@%s%s%s%s+%s%s%s%s"%(name,nl,body,nl,rnd.choice(["",name]),nl,q,nl))
        else:
            w=rnd.choice([1000,20,7]); body=nl.join(s[j:j+w] for j in range(0,len(s),w))
            recs.append(">%s%s%s%s"%(name,nl,body,nl if body else ""))
            if rnd.random()<0.1: recs.append(nl)
    text="".join(recs)
    if rnd.random()<0.15: text=text.rstrip("\r\n")
    p=os.path.join(tmp,"x"); open(p,"w",newline="").write(text)
    extra=rnd.choice([[],[],["-5","2"],["-3","3"],["-5","4","-3","1"]])
    fmt=["-q"] if fq else ["-f"]
    r=ref(fmt+extra+["-U",p])
    if r=="TIMEOUT": continue
    for thr,env in (("1",None),("3",None),("2",{"CF_INGEST_BLOCK":"4096"}),("2",{"CF_INGEST_BLOCK":"4096","CF_INGEST_STREAM":"1"})):
        m=mine(fmt+extra+["-p",thr,"-U",p],env)
        if m!=r:
            bad+=1; print("DIFF",it-1,fmt,extra,"p"+thr,"\n ref ",r if r=="ERR" else r[:12],"\n mine",m if m=="ERR" else m[:12]); 
            import shutil; shutil.copy(p,"/tmp/bad_ingest_%d_%s"%(it-1,"fq" if fq else "fa")); break
print("iterations",it,"bad",bad)
