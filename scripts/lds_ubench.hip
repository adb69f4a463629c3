// Micro-benchmark: LDS atomic / plain access rates per CU on MI355X (random addresses, 16 KiB-128 KiB tables).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t xs(uint32_t &s){ s^=s<<13; s^=s>>17; s^=s<<5; return s; }
template<int MODE> __global__ __launch_bounds__(256) void k(int iters, uint32_t mask, unsigned long long* sink){
  extern __shared__ uint64_t lds[];
  uint32_t* l32=(uint32_t*)lds;
  for(uint32_t i=threadIdx.x;i<=mask;i+=256) lds[i]=0;
  __syncthreads();
  uint32_t s=threadIdx.x*2654435761u+blockIdx.x*40503u+1; unsigned long long acc=0;
  for(int it=0;it<iters;++it){
    uint32_t a=xs(s)&mask;
    if(MODE==0) acc+=atomicAdd(&l32[a],1u);
    else if(MODE==1) atomicAdd(&l32[a],1u);
    else if(MODE==2) acc+=atomicCAS(&l32[a],0u,s|1u);
    else if(MODE==3) acc+=atomicCAS((unsigned long long*)&lds[a],0ULL,(unsigned long long)s|1ULL);
    else if(MODE==4) acc+=atomicAdd((unsigned long long*)&lds[a],1ULL);
    else if(MODE==5) { acc+=lds[a]; }
    else if(MODE==6) { lds[a]=s; }
    else if(MODE==7) atomicAdd((unsigned long long*)&lds[a],1ULL);
  }
  if(acc==0x1234567) *sink=acc;
}
int main(){
  unsigned long long* sink; (void)hipMalloc(&sink,8);
  hipEvent_t a,b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const char* names[8]={"add_rtn_u32","add_u32_noret","cas_rtn_b32","cas_rtn_b64","add_rtn_u64","read_b64","write_b64","add_u64_noret"};
  const int iters=4096, grid=256*4;
  for(int kb: {16, 32}){
    uint32_t mask=(kb*1024/8)-1; size_t lds=(size_t)kb*1024;
    for(int mode=0;mode<8;++mode){
      for(int rep=0;rep<2;++rep){
      (void)hipEventRecord(a);
      switch(mode){
        case 0:k<0><<<grid,256,lds>>>(iters,mask,sink);break; case 1:k<1><<<grid,256,lds>>>(iters,mask,sink);break;
        case 2:k<2><<<grid,256,lds>>>(iters,mask,sink);break; case 3:k<3><<<grid,256,lds>>>(iters,mask,sink);break;
        case 4:k<4><<<grid,256,lds>>>(iters,mask,sink);break; case 5:k<5><<<grid,256,lds>>>(iters,mask,sink);break;
        case 6:k<6><<<grid,256,lds>>>(iters,mask,sink);break; default:k<7><<<grid,256,lds>>>(iters,mask,sink);}
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);}
      float ms; (void)hipEventElapsedTime(&ms,a,b);
      double ops=(double)grid*256*iters;
      printf("lds=%3d KiB %-14s %8.3f ms  %8.2f Gops/s chip  %6.3f lane-ops/clk/CU (2.4GHz)\n",kb,names[mode],ms,ops/ms/1e6, ops/ms/1e6*1e9/256/2.4e9);
    }
  }
  return 0;
}
