import numpy as np
N=512; G=64; EPT=8
W=lambda e,s: np.exp(s*2j*np.pi*e/N)
def dft8(A,s):
    B=np.zeros_like(A)
    for q in range(8):
        for a in range(8): B[q]+=A[a]*np.exp(s*2j*np.pi*a*q/8)
    return B
def swap3(j): return ((j&7)<<3)|(j>>3)
def inverse(Z,s=+1):
    A=np.array([[Z[64*t+j] for j in range(G)] for t in range(8)])   # j = wave*8 + lane3
    A=dft8(A,s)
    for t in range(8):
        for j in range(G): A[t][j]*=W(t*j,s)
    # X1: (wave=a, reg=q) -> (wave=q, reg=a); lanes (m) stay
    B=np.zeros_like(A)
    for q in range(8):
        for a in range(8):
            for m in range(8): B[a][q*8+m]=A[q][a*8+m]
    A=dft8(B,s)
    for t in range(8):
        for j in range(G): A[t][j]*=W(8*t*(j&7),s)
    # X2: swap register index with lane3
    B=np.zeros_like(A)
    for t in range(8):
        for w in range(8):
            for m in range(8): B[m][w*8+t]=A[t][w*8+m]
    A=dft8(B,s)
    return A    # register t of thread j holds element swap3(j) + 64 t
def forward(P,s=-1):
    A=dft8(P.copy(),s)
    B=np.zeros_like(A)
    for t in range(8):
        for w in range(8):
            for m in range(8): B[m][w*8+t]=A[t][w*8+m]
    A=B
    for t in range(8):
        for j in range(G): A[t][j]*=W(8*t*(j&7),s)
    A=dft8(A,s)
    B=np.zeros_like(A)
    for q in range(8):          # transpose of X1: (wave=q, reg=a) -> (wave=a, reg=q)
        for a in range(8):
            for m in range(8): B[q][a*8+m]=A[a][q*8+m]
    A=B
    for t in range(8):
        for j in range(G): A[t][j]*=W(t*j,s)
    return dft8(A,s)
rng=np.random.default_rng(1)
Z=rng.standard_normal(N)+1j*rng.standard_normal(N)
out=inverse(Z); ref=np.fft.ifft(Z)*N
got=np.zeros(N,complex)
for t in range(8):
    for j in range(G): got[swap3(j)+64*t]=out[t][j]
print("inverse err",np.abs(got-ref).max()/np.abs(ref).max())
p=rng.standard_normal(N)+1j*rng.standard_normal(N)
P=np.array([[p[swap3(j)+64*t] for j in range(G)] for t in range(8)])
F=forward(P); reff=np.fft.fft(p)
gotf=np.array([F[n>>6][n&63] for n in range(N)])
print("forward err",np.abs(gotf-reff).max()/np.abs(reff).max())
