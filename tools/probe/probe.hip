// Hardware-semantics probe for gfx950: checks the lane maps this repo's kernels assume
// (MFMA operand/accumulator layouts, ds_read_b64_tr_b16 gather, global_load_lds placement).
// Build: hipcc --offload-arch=gfx950 -O2 probe.hip -o probe ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

static unsigned short f2bf(float f){ unsigned u; memcpy(&u,&f,4); return (unsigned short)((u + 0x7FFF + ((u>>16)&1))>>16); }

// ---- 1. mfma 16x16x32: A[16][32], B[32][16] (B given as Bt[n][k]) ----
__global__ void k_mfma16(const __bf16* A, const __bf16* Bt, float* C){
  int l = threadIdx.x;
  bf8 a, b;
  for(int j=0;j<8;j++){ a[j] = A[(l&15)*32 + 8*(l>>4)+j]; b[j] = Bt[(l&15)*32 + 8*(l>>4)+j]; }
  f4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,c,0,0,0);
  for(int r=0;r<4;r++) C[((l>>4)*4+r)*16 + (l&15)] = c[r];
}
// ---- 2. mfma 32x32x16: A[32][16], Bt[32][16] ----
__global__ void k_mfma32(const __bf16* A, const __bf16* Bt, float* C){
  int l = threadIdx.x;
  bf8 a, b;
  for(int j=0;j<8;j++){ a[j] = A[(l&31)*16 + 8*(l>>5)+j]; b[j] = Bt[(l&31)*16 + 8*(l>>5)+j]; }
  f16v c; for(int i=0;i<16;i++) c[i]=0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,c,0,0,0);
  for(int r=0;r<16;r++) C[((r&3)+8*(r>>2)+4*(l>>5))*32 + (l&31)] = c[r];
}
// ---- 3. tr read: LDS tile [rows][pitch] of shorts, value = row*256+col. lane 16g+4q+p supplies
//          address of (row r0+rowsel(g,q), col c0 + 4p). dump 4 shorts per lane.
__global__ void k_tr(short* out, int pitch){
  __shared__ __attribute__((aligned(16))) short lds[64*64];
  int l = threadIdx.x;
  for(int i=l;i<64*64;i+=64) lds[i] = (short)((i/pitch)*256 + (i%pitch));
  __syncthreads();
  int g = l>>4, q=(l>>2)&3, p=l&3;
  int row = 8*g + q, col = 4*p;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + row*pitch + col));
  for(int j=0;j<4;j++) out[l*4+j]=v[j];
}
// ---- 4. glds: each lane loads 16B from src + perm(lane)*16 into LDS base; dump LDS ----
__global__ void k_glds(const int* src, int* out){
  __shared__ __attribute__((aligned(16))) int lds[64*4*2];
  int l = threadIdx.x;
  for(int i=l;i<512;i+=64) lds[i] = -1;
  __syncthreads();
  int srcchunk = (l*7)&63;   // a permutation of 0..63
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + srcchunk*4),
        (void __attribute__((address_space(3)))*)(lds), 16, 0, 0);
  // second instruction with immediate offset 1024 bytes
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + l*4),
        (void __attribute__((address_space(3)))*)(lds), 16, 1024, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for(int i=l;i<512;i+=64) out[i]=lds[i];
}
// ---- 5. accumulator-as-B-operand: X = A1*B1 (32x32, K=16) then Y = A2 * X (A2 is 32x32, K=32 over X rows)
__global__ void k_acc_as_b(const __bf16* A1, const __bf16* B1t, const __bf16* A2, float* Y){
  int l = threadIdx.x; int r = l&31, h = l>>5;
  bf8 a, b;
  for(int j=0;j<8;j++){ a[j] = A1[r*16 + 8*h+j]; b[j] = B1t[r*16 + 8*h+j]; }
  f16v x; for(int i=0;i<16;i++) x[i]=0;
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,x,0,0,0);   // X[row][col=lane&31]
  f16v y; for(int i=0;i<16;i++) y[i]=0;
  for(int s=0;s<2;s++){
    bf8 xb, a2;
    for(int j=0;j<8;j++){
      xb[j] = (__bf16)x[8*s+j];
      int krow = 16*s + 8*(j>>2) + 4*h + (j&3);     // row of X this element is
      a2[j] = A2[r*32 + krow];                      // A2[row r][k = krow]
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0,0,0);
  }
  for(int q=0;q<16;q++) Y[((q&3)+8*(q>>2)+4*h)*32 + r] = y[q];
}

int main(){
  int fails=0;
  // 1
  {
    std::vector<unsigned short> A(16*32), Bt(16*32); std::vector<float> Af(16*32), Bf(16*32);
    for(int i=0;i<16;i++)for(int k=0;k<32;k++){ float v=(float)((i*3+k*5)%7-3); Af[i*32+k]=v; A[i*32+k]=f2bf(v);}
    for(int n=0;n<16;n++)for(int k=0;k<32;k++){ float v=(float)((n*7+k*2+n*k)%5-2); Bf[n*32+k]=v; Bt[n*32+k]=f2bf(v);}
    void *dA,*dB; float* dC; CK(hipMalloc(&dA,1024)); CK(hipMalloc(&dB,1024)); CK(hipMalloc(&dC,1024));
    CK(hipMemcpy(dA,A.data(),1024,hipMemcpyHostToDevice)); CK(hipMemcpy(dB,Bt.data(),1024,hipMemcpyHostToDevice));
    k_mfma16<<<1,64>>>((__bf16*)dA,(__bf16*)dB,dC); CK(hipDeviceSynchronize());
    float C[256]; CK(hipMemcpy(C,dC,1024,hipMemcpyDeviceToHost));
    int bad=0; for(int i=0;i<16;i++)for(int n=0;n<16;n++){ float s=0; for(int k=0;k<32;k++) s+=Af[i*32+k]*Bf[n*32+k]; if(fabs(s-C[i*16+n])>1e-3) bad++; }
    printf("[1] mfma16x16x32 layout: %s (bad=%d)\n", bad?"FAIL":"PASS", bad); fails+=bad!=0;
  }
  // 2
  {
    std::vector<unsigned short> A(32*16), Bt(32*16); std::vector<float> Af(32*16), Bf(32*16);
    for(int i=0;i<32;i++)for(int k=0;k<16;k++){ float v=(float)((i*3+k*5)%7-3); Af[i*16+k]=v; A[i*16+k]=f2bf(v);}
    for(int n=0;n<32;n++)for(int k=0;k<16;k++){ float v=(float)((n*7+k*2+n*k)%5-2); Bf[n*16+k]=v; Bt[n*16+k]=f2bf(v);}
    void *dA,*dB; float* dC; CK(hipMalloc(&dA,1024)); CK(hipMalloc(&dB,1024)); CK(hipMalloc(&dC,4096));
    CK(hipMemcpy(dA,A.data(),1024,hipMemcpyHostToDevice)); CK(hipMemcpy(dB,Bt.data(),1024,hipMemcpyHostToDevice));
    k_mfma32<<<1,64>>>((__bf16*)dA,(__bf16*)dB,dC); CK(hipDeviceSynchronize());
    float C[1024]; CK(hipMemcpy(C,dC,4096,hipMemcpyDeviceToHost));
    int bad=0; for(int i=0;i<32;i++)for(int n=0;n<32;n++){ float s=0; for(int k=0;k<16;k++) s+=Af[i*16+k]*Bf[n*16+k]; if(fabs(s-C[i*32+n])>1e-3) bad++; }
    printf("[2] mfma32x32x16 layout: %s (bad=%d)\n", bad?"FAIL":"PASS", bad); fails+=bad!=0;
  }
  // 3
  for(int pitch : {64, 16, 40}){
    short* d; CK(hipMalloc(&d,64*4*2));
    k_tr<<<1,64>>>(d,pitch); CK(hipDeviceSynchronize());
    short o[256]; CK(hipMemcpy(o,d,512,hipMemcpyDeviceToHost));
    // expectation: lane 16g+i receives column i (c0=0..15) of rows 8g+0..3 : element q = row 8g+q, col i
    int bad=0;
    for(int l=0;l<64;l++){ int g=l>>4,i=l&15; for(int q=0;q<4;q++){ int exp=(8*g+q)*256 + i; if(o[l*4+q]!=(short)exp) bad++; } }
    printf("[3] ds_read_tr16_b64 (pitch %d shorts): %s (bad=%d)\n", pitch, bad?"FAIL":"PASS", bad); fails+=bad!=0;
    if(bad){ for(int l=0;l<64;l++){ printf("  lane %2d:",l); for(int q=0;q<4;q++) printf(" (r%d,c%d)", ((unsigned short)o[l*4+q])>>8, o[l*4+q]&255); printf("\n"); } }
  }
  // 4
  {
    int h[256]; for(int i=0;i<256;i++) h[i]=i; int *ds,*dout; CK(hipMalloc(&ds,1024)); CK(hipMalloc(&dout,2048));
    CK(hipMemcpy(ds,h,1024,hipMemcpyHostToDevice));
    k_glds<<<1,64>>>(ds,dout); CK(hipDeviceSynchronize());
    int o[512]; CK(hipMemcpy(o,dout,2048,hipMemcpyDeviceToHost));
    int bad=0;
    for(int l=0;l<64;l++) for(int j=0;j<4;j++){ int exp=((l*7)&63)*4+j; if(o[l*4+j]!=exp) bad++; if(o[256+l*4+j]!=l*4+j) bad++; }
    printf("[4] global_load_lds placement (lane-linear dst, per-lane src, imm offset): %s (bad=%d)\n", bad?"FAIL":"PASS", bad); fails+=bad!=0;
    if(bad){ for(int i=0;i<512;i++){ printf("%d ",o[i]); if(i%16==15)printf("\n"); } }
  }
  // 5
  {
    std::vector<unsigned short> A1(32*16), B1t(32*16), A2(32*32); std::vector<float> A1f(32*16), B1f(32*16), A2f(32*32);
    for(int i=0;i<32;i++)for(int k=0;k<16;k++){ float v=(float)((i*3+k*5)%3-1); A1f[i*16+k]=v; A1[i*16+k]=f2bf(v);}
    for(int n=0;n<32;n++)for(int k=0;k<16;k++){ float v=(float)((n*7+k*2+n*k)%3-1); B1f[n*16+k]=v; B1t[n*16+k]=f2bf(v);}
    for(int i=0;i<32;i++)for(int k=0;k<32;k++){ float v=(float)((i*5+k*3+i*k)%5-2); A2f[i*32+k]=v; A2[i*32+k]=f2bf(v);}
    void *d1,*d2,*d3; float* dY; CK(hipMalloc(&d1,1024)); CK(hipMalloc(&d2,1024)); CK(hipMalloc(&d3,2048)); CK(hipMalloc(&dY,4096));
    CK(hipMemcpy(d1,A1.data(),1024,hipMemcpyHostToDevice)); CK(hipMemcpy(d2,B1t.data(),1024,hipMemcpyHostToDevice)); CK(hipMemcpy(d3,A2.data(),2048,hipMemcpyHostToDevice));
    k_acc_as_b<<<1,64>>>((__bf16*)d1,(__bf16*)d2,(__bf16*)d3,dY); CK(hipDeviceSynchronize());
    float Y[1024]; CK(hipMemcpy(Y,dY,4096,hipMemcpyDeviceToHost));
    std::vector<float> X(32*32);
    for(int i=0;i<32;i++)for(int n=0;n<32;n++){ float s=0; for(int k=0;k<16;k++) s+=A1f[i*16+k]*B1f[n*16+k]; X[i*32+n]=s; }
    int bad=0; for(int i=0;i<32;i++)for(int n=0;n<32;n++){ float s=0; for(int k=0;k<32;k++) s+=A2f[i*32+k]*X[k*32+n]; if(fabs(s-Y[i*32+n])>1e-2) bad++; }
    printf("[5] accumulator-as-B-operand (Y=A2*X): %s (bad=%d)\n", bad?"FAIL":"PASS", bad); fails+=bad!=0;
  }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device: %s CUs=%d clock=%d MHz mem=%.1f GB LDS/block=%zu\n", p.name, p.multiProcessorCount, p.clockRate/1000, p.totalGlobalMem/1e9, p.sharedMemPerBlock);
  printf("probe %s\n", fails?"HAS FAILURES":"ALL PASS");
  return 0;
}
