#!/usr/bin/env python3
"""Least-squares fits of the two-branch error function behind msn::gelu_both (csrc/msn_common.h), checked in fp32 arithmetic
against scipy: erf(z) = z P(z^2) on |z| <= 1, erfc(z) = exp(-z^2) t Q(t), t = 1 / (1 + p z), beyond."""
import numpy as np
from scipy import special
np.set_printoptions(precision=10, linewidth=200)
f32=np.float32
# ---- branch S: erf(z) = z * P(z^2) on |z| <= ZS
ZS=1.0
def fitS(deg):
    z=np.cos(np.linspace(0,np.pi,4001))*0.5*ZS+0.5*ZS   # chebyshev nodes in [0,ZS]
    z=z[z>1e-6]
    s=z*z
    y=special.erf(z)/z
    A=np.stack([s**k for k in range(deg)],axis=1)
    c,*_=np.linalg.lstsq(A,y,rcond=None)
    return c
def evalS(c,z):
    z=z.astype(f32); s=(z*z).astype(f32)
    acc=np.full_like(z,f32(c[-1]))
    for k in c[-2::-1]:
        acc=(acc*s+f32(k)).astype(f32)
    return (acc*z).astype(f32)
for deg in (6,7,8):
    c=fitS(deg); zt=np.linspace(0,ZS,200001)
    e=np.abs(evalS(c,zt).astype(np.float64)-special.erf(zt)).max()
    print("S deg",deg,"max abs err",e)
cS=fitS(7)
print("cS =",[float(f32(x)) for x in cS])
# ---- branch L: erfc(z) = exp(-z^2) * t * Q(t), t = 1/(1+p z), z >= ZS
def fitL(p,deg,zmax=7.0):
    z=np.linspace(ZS*0.98,zmax,20001)
    t=1/(1+p*z)
    y=special.erfcx(z)/t
    A=np.stack([t**k for k in range(deg)],axis=1)
    w=np.exp(-z*z)*t+1e-6
    c,*_=np.linalg.lstsq(A*w[:,None],y*w,rcond=None)
    return c
def evalL(c,p,z):
    z=z.astype(f32)
    t=(f32(1)/(f32(1)+f32(p)*z)).astype(f32)
    acc=np.full_like(z,f32(c[-1]))
    for k in c[-2::-1]:
        acc=(acc*t+f32(k)).astype(f32)
    e=np.exp(-(z*z).astype(f32)).astype(f32)
    return (acc*t).astype(f32)*e
best=None
for p in (0.3,0.4,0.47047,0.5,0.6,0.8,1.0):
    for deg in (6,7,8,9):
        c=fitL(p,deg); zt=np.linspace(ZS,7,200001)
        e=np.abs(evalL(c,p,zt).astype(np.float64)-special.erfc(zt)).max()
        if best is None or e<best[0]: best=(e,p,deg,c)
        print("L p",p,"deg",deg,"max abs err erfc",e)
print(best[:3]); print("cL =",[float(f32(x)) for x in best[3]])
