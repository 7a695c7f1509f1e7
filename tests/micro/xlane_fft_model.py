import numpy as np
N=1024; G=128; EPT=8
W=lambda e,s: np.exp(s*2j*np.pi*e/N)
def lanebits(j):
    w=j>>6; l=j&63
    return w,[(l>>b)&1 for b in range(6)]
def jprime(q,m):
    # lane id after the LDS exchange: w = q2, (l5 l4 l3 l2) = m, (l1 l0) = (q1 q0)
    return ((q>>2)&1)*64 + (m<<2 | (q&3))
def qm_of(jp):
    w,l=jp>>6,jp&63
    return (w<<2)|(l&3), l>>2
def swap_reg_lane(A, regbit, lanebit):
    # exchange register-index bit `regbit` with lane-index bit `lanebit` (lane id includes wave bit 6)
    B=A.copy()
    for t in range(EPT):
        for j in range(G):
            tb=(t>>regbit)&1; lb=(j>>lanebit)&1
            if tb!=lb:
                t2=t^(1<<regbit); j2=j^(1<<lanebit)
                B[t2][j2]=A[t][j]
    return B
def dft_regs(A, bits, s):
    # DFT over the register bits listed (MSB first) ; result digit stored in the same bits (MSB first)
    R=1<<len(bits)
    B=np.zeros_like(A)
    for t in range(EPT):
        # digit value of t
        q=0
        for b in bits: q=(q<<1)|((t>>b)&1)
        for a in range(R):
            ta=t
            for i,b in enumerate(bits):
                bit=(a>>(len(bits)-1-i))&1
                ta=(ta&~(1<<b))|(bit<<b)
            B[t]+=A[ta]*np.exp(s*2j*np.pi*a*q/R)
    return B
def digit(t,bits):
    q=0
    for b in bits: q=(q<<1)|((t>>b)&1)
    return q
def inverse(Z, s=+1):
    A=np.array([[Z[128*t+j] for j in range(G)] for t in range(EPT)])
    # stage 1
    A=dft_regs(A,[2,1,0],s)
    for t in range(EPT):
        for j in range(G): A[t][j]*=W(t*j,s)
    # LDS exchange
    B=np.zeros_like(A)
    for jp in range(G):
        q,m=qm_of(jp)
        for a in range(EPT): B[a][jp]=A[q][16*a+m]
    A=B
    # stage 2
    A=dft_regs(A,[2,1,0],s)
    for t in range(EPT):
        for jp in range(G):
            q,m=qm_of(jp); A[t][jp]*=W(8*t*m,s)
    # stage 3: swap t2<->l5, t1<->l4
    A=swap_reg_lane(A,2,5); A=swap_reg_lane(A,1,4)
    A=dft_regs(A,[2,1],s)
    for t in range(EPT):
        q3=digit(t,[2,1])
        for jp in range(G):
            c=(jp>>2)&3; A[t][jp]*=W(64*q3*c,s)
    # stage 4: swap t2<->l3, t1<->l2
    A=swap_reg_lane(A,2,3); A=swap_reg_lane(A,1,2)
    A=dft_regs(A,[2,1],s)
    return A
def kmap():
    K=np.zeros((EPT,G),dtype=int)
    for t in range(EPT):
        for j in range(G):
            w=j>>6; l=j&63; lb=lambda b:(l>>b)&1
            K[t][j]=((w<<2)|(lb(1)<<1)|lb(0)) + 8*((lb(5)<<2)|(lb(4)<<1)|(t&1)) + 64*((lb(3)<<1)|lb(2)) + 256*(((t>>2)&1)<<1|((t>>1)&1))
    return K
def forward(P, s=-1):
    A=P.copy()   # layout pi
    A=dft_regs(A,[2,1],s)
    A=swap_reg_lane(A,1,2); A=swap_reg_lane(A,2,3)
    for t in range(EPT):
        q3=digit(t,[2,1])
        for jp in range(G):
            c=(jp>>2)&3; A[t][jp]*=W(64*q3*c,s)
    A=dft_regs(A,[2,1],s)
    A=swap_reg_lane(A,1,4); A=swap_reg_lane(A,2,5)
    for t in range(EPT):
        for jp in range(G):
            q,m=qm_of(jp); A[t][jp]*=W(8*t*m,s)
    A=dft_regs(A,[2,1,0],s)
    B=np.zeros_like(A)
    for jp in range(G):
        q,m=qm_of(jp)
        for a in range(EPT): B[q][16*a+m]=A[a][jp]
    A=B
    for t in range(EPT):
        for j in range(G): A[t][j]*=W(t*j,s)
    A=dft_regs(A,[2,1,0],s)
    return A
rng=np.random.default_rng(0)
Z=rng.standard_normal(N)+1j*rng.standard_normal(N)
out=inverse(Z); K=kmap()
ref=np.fft.ifft(Z)*N
got=np.zeros(N,dtype=complex)
for t in range(EPT):
    for j in range(G): got[K[t][j]]=out[t][j]
print("inverse err",np.abs(got-ref).max()/np.abs(ref).max(), "perm ok", sorted(K.flatten().tolist())==list(range(N)))
p=rng.standard_normal(N)+1j*rng.standard_normal(N)
P=np.array([[p[K[t][j]] for j in range(G)] for t in range(EPT)])
F=forward(P)
reff=np.fft.fft(p)
gotf=np.array([F[n>>7][n&127] for n in range(N)])
print("forward err",np.abs(gotf-reff).max()/np.abs(reff).max())
