"""Integer model of secp256k1_voi_amd/csrc/modinv30.h: modular inversion mod the group order by the
safegcd division steps on 9 signed 30-bit limbs, with the 32-/64-bit wrap-around of the device code
made explicit.  Checks the limb schedule (20 rounds of 30 steps, the exact-division updates, the
final normalisation) against pow(x, -1, n) on edge values and random scalars."""
import random
N=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
M30=(1<<30)-1
def s32(x):
    x&=0xffffffff
    return x-(1<<32) if x&0x80000000 else x
def s64(x):
    x&=(1<<64)-1
    return x-(1<<64) if x>>63 else x
def to30(x):
    return [(x>>(30*i))&M30 for i in range(9)]
def val(v): return sum(x<<(30*i) for i,x in enumerate(v))
MOD=to30(N)
NINV30=pow(N,-1,1<<30)
def divsteps30(zeta,f0,g0):
    u,v,q,r=1,0,0,1
    f,g=f0&0xffffffff,g0&0xffffffff
    for i in range(30):
        c1=s32(zeta)>>31 & 0xffffffff   # all ones if negative
        c2=(-(g&1))&0xffffffff
        x=((f^c1)-c1)&0xffffffff; y=((u^c1)-c1)&0xffffffff; z=((v^c1)-c1)&0xffffffff
        g=(g+(x&c2))&0xffffffff; q=(q+(y&c2))&0xffffffff; r=(r+(z&c2))&0xffffffff
        c1&=c2
        zeta=s32(((zeta&0xffffffff)^c1)-1)
        f=(f+(g&c1))&0xffffffff; u=(u+(q&c1))&0xffffffff; v=(v+(r&c1))&0xffffffff
        g>>=1; u=(u<<1)&0xffffffff; v=(v<<1)&0xffffffff
    return zeta,(s32(u),s32(v),s32(q),s32(r))
def update_fg(f,g,t):
    u,v,q,r=t
    cf=s64(u*f[0]+v*g[0]); cg=s64(q*f[0]+r*g[0])
    assert cf&M30==0 and cg&M30==0
    cf>>=30; cg>>=30
    for i in range(1,9):
        cf=s64(cf+u*f[i]+v*g[i]); cg=s64(cg+q*f[i]+r*g[i])
        f[i-1]=cf&M30; cf>>=30; g[i-1]=cg&M30; cg>>=30
    f[8]=s32(cf); g[8]=s32(cg)
def update_de(d,e,t):
    u,v,q,r=t
    sd=d[8]>>31; se=e[8]>>31   # python ints: -1 or 0
    md=(u&sd)+(v&se); me=(q&sd)+(r&se)
    di,ei=d[0],e[0]
    cd=s64(u*di+v*ei); ce=s64(q*di+r*ei)
    md-=(NINV30*(cd&0xffffffff)+md)&M30
    me-=(NINV30*(ce&0xffffffff)+me)&M30
    cd=s64(cd+MOD[0]*md); ce=s64(ce+MOD[0]*me)
    assert cd&M30==0 and ce&M30==0
    cd>>=30; ce>>=30
    for i in range(1,9):
        di,ei=d[i],e[i]
        cd=s64(cd+u*di+v*ei); ce=s64(ce+q*di+r*ei)
        cd=s64(cd+MOD[i]*md); ce=s64(ce+MOD[i]*me)
        d[i-1]=cd&M30; cd>>=30; e[i-1]=ce&M30; ce>>=30
    d[8]=s32(cd); e[8]=s32(ce)
def normalize(r,sign):
    r=list(r)
    cond_add=r[8]>>31
    r=[r[i]+(MOD[i]&cond_add) for i in range(9)]
    cn=sign>>31
    r=[(x^cn)-cn for x in r]
    for i in range(8):
        r[i+1]+=r[i]>>30; r[i]&=M30
    cond_add=r[8]>>31
    r=[r[i]+(MOD[i]&cond_add) for i in range(9)]
    for i in range(8):
        r[i+1]+=r[i]>>30; r[i]&=M30
    return r
def modinv(x):
    d=[0]*9; e=[1]+[0]*8; f=list(MOD); g=to30(x)
    zeta=-1
    for it in range(20):
        zeta,t=divsteps30(zeta,f[0],g[0])
        update_de(d,e,t)
        update_fg(f,g,t)
    assert all(x==0 for x in g), g
    return val(normalize(d,f[8]))


def test_modinv_schedule_matches_pow():
    rnd = random.Random(5)
    xs = [0, 1, 2, N - 1, N - 2, (N + 1) // 2, 1 << 255, (1 << 128) - 1, (1 << 30) - 1, 1 << 30, M30 << 226]
    xs += [rnd.randrange(N) for _ in range(1500)]
    for x in xs:
        x %= N
        assert modinv(x) == (pow(x, -1, N) if x else 0)


def test_constants():
    assert val(MOD) == N and (N * NINV30) % (1 << 30) == 1
