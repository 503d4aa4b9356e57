"""CPU: the thread -> LDS image -> MFMA fragment -> accumulator -> output mapping of csrc/gemm_mfma.hip, restated with
numpy index arithmetic and checked against A @ B for all four operand layouts and both arithmetic modes.

What is restated (and must be kept in step with the kernel): the staging maps of `load_kmajor` / `load_mnmajor` /
`store_kmajor` / `store_mnmajor` (which element a thread holds and where it lands in the image), the image geometry
`Geo<PREC>` (K-major row stride 36 dwords in fp32 mode; 16 dwords with XOR-swizzled 16-byte chunks in bf16 mode;
136-dword unit rows), `frag()` (units 8s + 4h + {0..3} of a row), the operand
roles of `v_mfma_f32_32x32x16_bf16` / `v_mfma_f32_32x32x2_f32` (A[i = lane & 31][k = lane >> 5 ...], B[k][j = lane & 31]) and
their accumulator layout (column lane & 31, row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)), and the epilogue's row / column
of a register.  The hi / lo split is not part of it (tests/test_gemm_cpu.py); values are carried exactly."""
import numpy as np
import pytest

BM = BN = 128
MN = 136
IMG = 4608


def emu(M,N,K,alay,blay,prec, A_store, B_store, lda, ldb):
    BK = 32
    CH = 2
    KM = 16 if prec else 36
    swz = lambda r, chunk: (chunk ^ ((r >> 2) & 3)) << 2          # bf16 K-major: dword offset of a row's 16-byte chunk
    NS = 2 if prec else 4
    C = np.zeros((M,N))
    Af = A_store.ravel(); Bf = B_store.ravel()
    for m0 in range(0,M,BM):
      for n0 in range(0,N,BN):
        acc = np.zeros((256,2,2,16))
        for k0 in range(0,K,BK):
            kend=K
            imgs = {}
            for name,P,ld,lay,r0,R in (("A",Af,lda,alay,m0,M),("B",Bf,ldb,blay,n0,N)):
                img = np.zeros(IMG)   # store float values per 'unit' (fp32 mode) ; bf16 mode: store pairs as complex-ish -> use 2 arrays
                img2 = np.zeros((IMG,2))
                for tid in range(256):
                    if lay==0:
                        LPR = 8; RPI=64//LPR; NI=32//RPI
                        lane=tid&63; wave=tid>>6; kq=lane%LPR
                        for i in range(NI):
                            r=32*wave+lane//LPR+i*RPI; row=r0+r; k=k0+kq*4
                            v=[(P[row*ld+k+j] if (row<R and k+j<kend) else 0.0) for j in range(4)]
                            if prec:
                                pos=r*KM+(swz(r,kq>>1)|((kq&1)<<1))
                                img2[pos]=(v[0],v[1]); img2[pos+1]=(v[2],v[3])
                            else:
                                for j in range(4): img[r*KM+4*kq+j]=v[j]
                    else:
                        if prec:
                            for c in range(2):
                                item=tid+256*c; col=r0+(item&31)*4; kk=k0+(item>>5)*2
                                for j in range(4):
                                    v0 = P[kk*ld+col+j] if (kk<kend and col+j<R) else 0.0
                                    v1 = P[(kk+1)*ld+col+j] if (kk+1<kend and col+j<R) else 0.0
                                    img2[(item>>5)*MN+(item&31)*4+j]=(v0,v1)
                        else:
                            for i in range(4):
                                item=tid+256*i; col=r0+(item&31)*4; kk=k0+(item>>5)
                                for j in range(4):
                                    img[(item>>5)*MN+(item&31)*4+j] = P[kk*ld+col+j] if (kk<kend and col+j<R) else 0.0
                imgs[name]=(img,img2,lay)
            def frag(name,row,s,h):
                img,img2,lay=imgs[name]
                src = img2 if prec else img
                if lay==0 and prec: return [src[row*KM+swz(row,2*s+h)+j] for j in range(4)]
                if lay==0: return [src[row*KM+8*s+4*h+j] for j in range(4)]
                return [src[(8*s+4*h+j)*MN+row] for j in range(4)]
            # per-wave MFMA emulation: D[i][j] += sum over lanes' k
            for wave in range(4):
                wm=(wave>>1)*64; wn=(wave&1)*64
                for s in range(NS):
                    for i in range(2):
                        for j in range(2):
                            # gather operand matrices: first operand rows n (32) x kslots ; second kslots x cols m (32)
                            if prec:
                                Aop=np.zeros((32,16)); Bop=np.zeros((16,32))
                                for lane in range(64):
                                    l31=lane&31; h=lane>>5
                                    fb=frag("B",wn+32*j+l31,s,h); fa=frag("A",wm+32*i+l31,s,h)
                                    for d in range(4):
                                        for e in range(2):
                                            Aop[l31, 8*h+2*d+e]=fa[d][e]; Bop[8*h+2*d+e, l31]=fb[d][e]
                                D=Aop@Bop
                            else:
                                D=np.zeros((32,32))
                                for u in range(4):
                                    Aop=np.zeros((32,2)); Bop=np.zeros((2,32))
                                    for lane in range(64):
                                        l31=lane&31; h=lane>>5
                                        fb=frag("B",wn+32*j+l31,s,h); fa=frag("A",wm+32*i+l31,s,h)
                                        Aop[l31,h]=fa[u]; Bop[h,l31]=fb[u]
                                    D+=Aop@Bop
                            for lane in range(64):
                                for reg in range(16):
                                    col=lane&31; row=(reg&3)+8*(reg>>2)+4*(lane>>5)
                                    acc[wave*64+lane,i,j,reg]+=D[row,col]
        for tid in range(256):
            lane=tid&63; wave=tid>>6; wm=(wave>>1)*64; wn=(wave&1)*64; l31=lane&31; h=lane>>5
            for j in range(2):
                n=n0+wn+32*j+l31
                if n>=N: continue
                for i in range(2):
                    for r in range(16):
                        m=m0+wm+32*i+(r&3)+8*(r>>2)+4*h
                        if m<M: C[m,n]=acc[tid,i,j,r]
    return C


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "bf16x3"])
@pytest.mark.parametrize("alay", [0, 1], ids=["A_kmajor", "A_mnmajor"])
@pytest.mark.parametrize("blay", [0, 1], ids=["B_kmajor", "B_mnmajor"])
def test_restated_mapping_reproduces_the_product(prec, alay, blay):
    rng = np.random.default_rng(prec * 4 + alay * 2 + blay)
    M, N, K = 130, 70, 45                          # ragged: two row tiles, one column tile, a k tail
    A = rng.uniform(-1, 1, (M, K)); B = rng.uniform(-1, 1, (K, N))
    As = A if alay == 0 else np.ascontiguousarray(A.T)
    Bs = np.ascontiguousarray(B.T) if blay == 0 else B
    C = emu(M, N, K, alay, blay, prec, As, Bs, As.shape[1], Bs.shape[1])
    assert np.abs(C - A @ B).max() < 1e-12
