import sys, os, math
sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden'); sys.path.insert(0,'.')
import numpy as np
import make_stat_golden as mg, stat_common as sc
from volume_renderer_amd import Backend, MODE_FAST, MODE_EXACT_SKIP
fx=np.load('tests/golden/stat_fixtures.npz')
b=Backend(0); b.set_density(mg.cloud())
for mode in (MODE_FAST, MODE_EXACT_SKIP):
  b.set_mode(mode)
  for name,integ,regime in mg.CASES:
    o=mg.options(integ,regime)
    fr=[]
    for k in range(16):
        b.set_seed(0xFACE0000+977*k); fr.append(b.render(o,integ).astype(np.float64))
    fr=np.stack(fr); m=fr.mean(0); se=fr.std(0,ddof=1)/4
    rm=fx[name+'_mean'].astype(np.float64); rse=fx[name+'_stderr'].astype(np.float64)
    den=np.sqrt(se**2+rse**2+(1e-5*rm)**2); z=(m-rm)/den
    print(mode,name,'z<3 %.4f z<4 %.4f'%(np.mean(abs(z)<3),np.mean(abs(z)<4)),'gpu>ref among |z|>3: %d of %d'%(np.sum(z>3),np.sum(abs(z)>3)),'mean z %.3f std z %.3f'%(z.mean(),z.std()), 'median ratio se_gpu*4/se_ref %.3f'%np.median((se*4)[rse>0]/rse[rse>0]), flush=True)
    # frame-vs-frame among gpu: frame0 vs mean of others
    d=(fr[0]-fr[1:].mean(0)); s=fr.std(0,ddof=1)*math.sqrt(1+1/15)
    zz=d/np.maximum(s,1e-30); ok=s>0
    print('   gpu frame0 vs other 15: z<3 %.4f'%np.mean(abs(zz[ok])<3))
