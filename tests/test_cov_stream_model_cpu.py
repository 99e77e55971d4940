"""Lane-level model of the streaming coverage reduction (ploidyfrost_amd/csrc/pf_cov_stream.hpp: k_cov_stream4, kc4_window,
seg_scan_narrow) against a direct per-unitig computation of CDBG::readCov(const UnitigMap&) (reference src/CDBG.cpp:66-120:
sum of counts, min initialised 10000, a missing k-mer flagged).  Host-side executable specification of what the HIP kernel
does per lane -- four k-mers per lane, (pre, suf) aggregates, segmented wave scan, wave-uniform carry across super-rows,
plain stores for unitigs that begin and end inside a window, atomics otherwise -- including the invariant that a plain-stored
unitig receives no other contribution.  The kernel itself is checked on the GPU (tests/test_gpu_kernels.py)."""
import random

M=0xFFFFFFFF
SRW=8
def popc(x): return bin(x).count("1")
def run(lens_k, counts, u0, u1, n_waves=3):
    N=len(lens_k); kpre=[0]
    for l in lens_k: kpre.append(kpre[-1]+l)
    nk=kpre[-1]; n_row=nk//64+1
    khead=[0]*(n_row+8); krow=[N-1]*(n_row+8); r=0
    for u in range(N):
        khead[kpre[u]>>6]|=1<<(kpre[u]&63)
        while r<n_row and r*64<kpre[u+1]: krow[r]=u; r+=1
    khead[nk>>6]|=1<<(nk&63)
    gcov=counts+[random.choice([M,5,99999])for _ in range((n_row+4)*64-nk)]
    n_out=u1-u0
    osum=[0]*n_out; omin=[10000]*n_out; omiss=[0]*n_out; plain=[0]*n_out; atom=[0]*n_out
    gb,ge=kpre[u0],kpre[u1]
    srb,sre=gb//256,(ge+255)//256
    n_win=(sre-srb+SRW-1)//SRW
    def emit(u,s,m,complete):
        o=u-u0
        if not (0<=o<n_out): return
        if complete:
            assert plain[o]==0 and atom[o]==0,(u,)
            plain[o]+=1; osum[o]=s; omin[o]=min(m,10000)
        else:
            assert plain[o]==0,(u,)
            atom[o]+=1; osum[o]+=s; omin[o]=min(omin[o],m)
    for wx in range(n_win):
        sr0=srb+wx*SRW
        csum=0;cmin=M;cstarted=False;ulast=None
        for j in range(SRW):
            sr=sr0+j
            if sr>=sre: break
            H=khead[sr*4:sr*4+4]; kb=krow[sr*4]; g0=sr*256
            edge=g0<gb or g0+256>ge
            hs=[];ubs=[];pre=[];suf=[];vs=[]
            for lane in range(64):
                w=lane>>4; sh=(lane&15)*4
                word=H[w]; h=(word>>sh)&0xF
                below=sum(popc(H[x]) for x in range(w))+popc(word&((2<<sh)-1))
                ub=kb+below-(H[0]&1)
                v=[gcov[g0+4*lane+i] for i in range(4)]
                if edge: v=[v[i] if gb<=g0+4*lane+i<ge else M for i in range(4)]
                x=[vi==M for vi in v]; s=[0 if x[i] else v[i] for i in range(4)]; m=v
                if any(x):
                    for i in range(4):
                        if x[i]:
                            o=ub+popc((h>>1)&((1<<i)-1))-u0
                            if 0<=o<n_out: omiss[o]=1
                ps=0;pm=M;ss=0;sm=M
                for i in range(4):
                    if (h&((2<<i)-1))==0: ps+=s[i]; pm=min(pm,m[i])
                    if (h>>(i+1))==0: ss+=s[i]; sm=min(sm,m[i])
                # interior complete segments
                if popc(h)>=2:
                    for i in range(3):
                        if (h>>i)&1 and (h>>(i+1)):
                            jn=i+1
                            while not (h>>jn)&1: jn+=1
                            emit(ub+popc((h>>1)&((1<<i)-1)), sum(s[i:jn]), min(m[i:jn]), True)
                hs.append(h);ubs.append(ub);pre.append((ps,pm));suf.append((ss,sm))
            F=sum((1<<l) for l in range(64) if hs[l])
            # segmented scan of suf with flags F (reference form; DPP form verified separately)
            V=[]
            for l in range(64):
                if hs[l] or l==0: V.append(suf[l])
                else: V.append((V[l-1][0]+suf[l][0], min(V[l-1][1],suf[l][1])))
            # carry-in
            for l in range(64):
                if (F&((2<<l)-1))==0: V[l]=(V[l][0]+csum,min(V[l][1],cmin))
            Vprev=[(csum,cmin)]+V[:63]
            for l in range(64):
                h=hs[l]
                if not h: continue
                if j==0 and l==0 and (h&1): continue
                E=(Vprev[l][0]+pre[l][0],min(Vprev[l][1],pre[l][1]))
                started=(F&((1<<l)-1))!=0 or cstarted
                emit(ubs[l]-(h&1),E[0],E[1],started)
            csum,cmin=V[63]; cstarted=cstarted or F!=0
            ulast=ubs[63]+popc(hs[63]>>1)
        emit(ulast,csum,cmin,False)
    return osum,omin,omiss


def direct(lens, counts, u0, u1):
    p = 0
    es, em, ex = [], [], []
    for u in range(len(lens)):
        seg = counts[p:p + lens[u]]
        p += lens[u]
        ok = [x for x in seg if x != M]
        es.append(sum(ok))
        em.append(min([10000] + ok))
        ex.append(int(any(x == M for x in seg)))
    return es[u0:u1], em[u0:u1], ex[u0:u1]


def test_window_reduction_model_matches_direct_computation():
    random.seed(5)
    for it in range(250):
        N = random.randint(1, 60)
        lens = [random.choice([1, 1, 1, 2, 3, 4, 5, 25, 47, 63, 64, 65, 128, 300, 700, 2500]) for _ in range(N)]
        counts = [random.choice([M, 20000, 3, 77]) if random.random() < 0.03 else random.randint(1, 60000) for _ in range(sum(lens))]
        u0 = random.randint(0, N - 1)
        u1 = random.randint(u0 + 1, N)
        if random.random() < 0.3:
            u0, u1 = 0, N
        assert run(lens, counts, u0, u1) == direct(lens, counts, u0, u1), (it, lens, u0, u1)


def test_one_unitig_longer_than_a_window_and_unitigs_of_one_kmer():
    random.seed(6)
    for lens in ([5000], [1] * 700, [1, 4000, 1, 1, 2047, 2048, 2049], [2048, 2048], [2047, 1, 2048]):
        counts = [random.randint(1, 60000) for _ in range(sum(lens))]
        assert run(lens, counts, 0, len(lens)) == direct(lens, counts, 0, len(lens))


K26 = (1 << 26) - 1


def scan_ref(F,s,mn,mx):
    rs=[];rm=[];rx=[]
    for l in range(64):
        a=0;b=M;c=0;j=l
        while True:
            a+=s[j];b=min(b,mn[j]);c=max(c,mx[j])
            if (F>>j)&1 or j==0: break
            j-=1
        rs.append(a);rm.append(b);rx.append(c)
    return rs,rm,rx
def scan_keys(F,s,mn,mx):
    d=[]
    for l in range(64):
        m=F&((2<<l)-1)
        d.append(l-(m.bit_length()-1) if m else 127)
    S=[l-d[l] if d[l]<64 else 0 for l in range(64)]
    P=[];acc=0
    for l in range(64): acc+=s[l];P.append(acc)
    vs=[P[l]-(P[S[l]-1] if S[l]>0 else 0) for l in range(64)]
    key=[((63-S[l])<<26)|min(mn[l],K26) for l in range(64)]
    out=[];cur=1<<40
    for l in range(64): cur=min(cur,key[l]);out.append(cur)
    vm=[(k&K26) for k in out]; vm=[M if v==K26 else v for v in vm]
    kx=[(S[l]<<26)|mx[l] for l in range(64)]
    ox=[];cur=0
    for l in range(64): cur=max(cur,kx[l]);ox.append(cur&K26)
    return vs,vm,ox


def test_narrow_scan_on_segment_tagged_keys_equals_segmented_scan():
    """seg_scan_narrow: sum = prefix-sum difference, min / max = unsegmented scans of keys carrying the segment's first lane."""
    random.seed(9)
    for it in range(3000):
        dens = random.choice([0, 0.02, 0.1, 0.5, 1.0])
        F = sum((1 << i) for i in range(64) if random.random() < dens)
        s = [random.randint(0, 4 * (1 << 20) - 4) for _ in range(64)]
        mn = [random.choice([M, random.randint(0, (1 << 20) - 1)]) for _ in range(64)]
        mx = [random.randint(0, (1 << 20) - 1) for _ in range(64)]
        assert scan_ref(F, s, mn, mx) == scan_keys(F, s, mn, mx), (it, hex(F))
