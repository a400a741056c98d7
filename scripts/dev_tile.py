import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
from gpsat_amd.engine import Engine
from gpsat_amd.local_experts import LocalSelector
np.random.seed(0)
N=100; noise_std=0.05
X=np.random.uniform(0.1,0.6,(N,)); y=np.sin(1/X)+noise_std*np.random.randn(N)
df=pd.DataFrame({'x':X,'y':y})
sel=LocalSelector(df,[{"col":"x","comp":"<=","val":0.15},{"col":"x","comp":">=","val":-0.15}])
eng=Engine(0)
for loc in (0.25,0.45):
    d=df.loc[sel.mask({"x":loc})]
    Xd=d[['x']].values.astype(np.float32); yd=d['y'].values.astype(np.float32)
    for kw in [dict(max_iter=200, ftol=1e-9), dict(max_iter=200, ftol=1e-12, max_ls=20)]:
        r=eng.fit_predict_batch(D=1,obs_off=[0,len(yd)],X=Xd,y=yd,pred_off=[0,0],Xs=np.zeros((0,1),np.float32),theta0=[[0.1,0.8,0.0025]],
                                trainable=[1,1,0],kernel="RBF",optimiser="lbfgs",want_grad=True,**kw)
        print(loc, kw, "status",r.status,"n_eval",r.n_eval,"theta",r.theta[0],"nll",r.nll[0],"grad",r.grad[0])
    for th in ([0.0321035,0.79929,0.0025],[0.0325,0.86,0.0025]):
        r=eng.fit_predict_batch(D=1,obs_off=[0,len(yd)],X=Xd,y=yd,pred_off=[0,0],Xs=np.zeros((0,1),np.float32),theta0=[th],kernel="RBF",optimiser="none",want_grad=True)
        print("   at",th,"nll",r.nll[0],"grad",r.grad[0])
