#!/usr/bin/env python3
"""The identity behind the zero-run skip of the PVQ leaf walk (og_celt_split.hpp, pvq_leaf_lane), checked in exact integers against
the step-by-step walk (cwrsi, src/celt.cpp:2545): with V(a) = U(a,k) + U(a,k+1) the dimensions n .. a+1 all decode to zero exactly
when V(n) - V(a) <= 2 i < V(n) + V(a), and skipping them subtracts (V(n) - V(a)) / 2 from the index.   python3 tools/pvq_zero_run.py"""
import random, functools, sys
sys.setrecursionlimit(10000)
@functools.lru_cache(None)
def U(n,k):
    # number of PVQ codewords ... U(n,k): as in libopus: U(0,k)=0 (k>0), U(n,0)=0?? use standard: U(n,k) = U(n-1,k)+U(n,k-1)+U(n-1,k-1), U(n,1)=1? 
    if k==0: return 0  # by convention used in cwrsi: U(n,0)=0 for n>0
    if n==0: return 0
    if n==1: return 1 if k>=1 else 0
    if k==1: return 1
    return U(n-1,k)+U(n,k-1)+U(n-1,k-1)
def V(n,k): return U(n,k)+U(n,k+1)
def cwrsi_ref(n,k,i):
    y=[]
    while n>2:
        p1=U(n,k+1); s=-(i>=p1); i-= p1 if s else 0
        p0=U(n,k)
        if p0<=i and s==0:
            i-=p0; y.append(0)
        else:
            k0=k
            # largest k' < k with U(n,k') <= i
            kk=k-1
            while U(n,kk)>i: kk-=1
            i-=U(n,kk); k=kk
            y.append((k0-k+s)^s)
        n-=1
    # n==2
    p=2*k+1; s=-(i>=p); i-= p if s else 0; k0=k; k=(i+1)>>1
    if k: i-=2*k-1
    y.append((k0-k+s)^s)
    s=-i; y.append((k+s)^s)
    return y
def P(n,a,k):  # sum_{t=a+1..n} U(t,k)
    v=U(n,k+1)-U(a,k+1)+U(n,k)-U(a,k)
    assert v%2==0
    return v//2
def cwrsi_skip(n,k,i):
    y=[]
    while n>2:
        if n>k and k>0:
            # zero run: positions t in [T,n] are zeros; find T by two monotone searches over t in [max(k+1,3), n]
            lo_t=max(k+1,3)
            # cond(t): zero at dimension t  <=> i >= P(n,t-1,k) and i < P(n,t,k)+U(t,k+1), valid for all t' in [t,n]
            def ok(t): return i>=P(n,t-1,k) and i< P(n,t,k)+U(t,k+1)
            # find smallest T in [lo_t, n+1] such that ok(t) for all t in [T,n]  (T=n+1: no zero)
            lo,hi=lo_t,n+1
            while lo<hi:
                mid=(lo+hi)//2
                if ok(mid): hi=mid
                else: lo=mid+1
            T=lo
            z=n-T+1
            if z>0:
                i-=P(n,T-1,k); y+= [0]*z; n=T-1
                continue
        # one ordinary step
        p1=U(n,k+1); s=-(i>=p1); i-= p1 if s else 0
        p0=U(n,k)
        if p0<=i and s==0:
            i-=p0; y.append(0)
        else:
            k0=k; kk=k-1
            while U(n,kk)>i: kk-=1
            i-=U(n,kk); k=kk
            y.append((k0-k+s)^s)
        n-=1
    p=2*k+1; s=-(i>=p); i-= p if s else 0; k0=k; k=(i+1)>>1
    if k: i-=2*k-1
    y.append((k0-k+s)^s)
    s=-i; y.append((k+s)^s)
    return y

def cwrsi_skip2(n,k,i, ratio=2):
    y=[]
    while n>2:
        if k==0:
            y += [0]*(n-2); n=2; break
        if k<=13 and n>ratio*k and n>3:
            Cn=V(n,k)
            lo=max(k+1,2); hi=n
            while lo<hi:
                mid=(lo+hi)//2
                Va=V(mid,k)
                if Cn-Va <= 2*i < Cn+Va: hi=mid
                else: lo=mid+1
            a=lo
            if a<n:
                i-=(Cn-V(a,k))//2; y+=[0]*(n-a); n=a
                if n<=2: break
        p1=U(n,k+1); s=-(i>=p1); i-= p1 if s else 0
        p0=U(n,k)
        if p0<=i and s==0:
            i-=p0; y.append(0)
        else:
            k0=k; kk=k-1
            while U(n,kk)>i: kk-=1
            i-=U(n,kk); k=kk
            y.append((k0-k+s)^s)
        n-=1
    p=2*k+1; s=-(i>=p); i-= p if s else 0; k0=k; k=(i+1)>>1
    if k: i-=2*k-1
    y.append((k0-k+s)^s)
    s=-i; y.append((k+s)^s)
    return y


def walk_steps(n, k, i, ratio):
    """Steps of the kernel's walk for one leaf: (iterations of its loop, of which with a bisection); ratio 0 = no skipping."""
    it = sk = 0
    while n > 2:
        if k == 0:
            break
        if ratio and k <= 13 and n > ratio * k and n > 3:
            sk += 1
            Cn = V(n, k); lo = max(k + 1, 2); hi = n
            while lo < hi:
                mid = (lo + hi) // 2; Va = V(mid, k)
                if Cn - Va <= 2 * i < Cn + Va: hi = mid
                else: lo = mid + 1
            if lo < n:
                i -= (Cn - V(lo, k)) // 2; n = lo
                if n <= 2: break
        it += 1
        p1 = U(n, k + 1); s = -(i >= p1); i -= p1 if s else 0
        p0 = U(n, k)
        if p0 <= i and s == 0: i -= p0
        else:
            kk = k - 1
            while U(n, kk) > i: kk -= 1
            i -= U(n, kk); k = kk
        n -= 1
    return it, sk

if __name__ == "__main__":
    random.seed(1)
    bad=0; tot=0
    for n in [3,4,5,6,8,9,11,12,16,18,22,24,32,36,44,48,64,72,96,144,176]:
        for k in [1,2,3,4,5,6,8,10,12,14,17,24,40,80,128]:
            if V(n,k)>=2**32: continue
            for _ in range(40):
                i=random.randrange(V(n,k))
                a=cwrsi_ref(n,k,i); b=cwrsi_skip(n,k,i)
                tot+=1
                if a!=b:
                    bad+=1
                    if bad<5: print("MISMATCH",n,k,i,a,b)
                assert sum(abs(v) for v in a)==k, (n,k,i,a)
    print(tot,"cases",bad,"bad")
    bad=0; tot=0; steps=0
    for n in [3,4,5,6,8,9,11,12,16,18,22,24,32,36,44,48,64,72,96,144,176]:
        for k in [1,2,3,4,5,6,8,10,12,13,14,17,24,40,80,128]:
            if V(n,k)>=2**32: continue
            for _ in range(60):
                i=random.randrange(V(n,k))
                a=cwrsi_ref(n,k,i)
                for r in (1,2,4):
                    b=cwrsi_skip2(n,k,i,r)
                    tot+=1
                    if a!=b:
                        bad+=1
                        if bad<5: print("MISMATCH2",n,k,i,r,a,b)
            # edge indices
            for i in (0,1,V(n,k)-1,V(n,k)-2,U(n,k+1),U(n,k+1)-1,U(n,k),U(n,k)-1):
                if 0<=i<V(n,k):
                    a=cwrsi_ref(n,k,i); b=cwrsi_skip2(n,k,i,1); tot+=1
                    if a!=b: bad+=1; print("EDGE",n,k,i)
    print(tot,"cases (skip2)",bad,"bad")
