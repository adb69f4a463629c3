// Micro-benchmark: random 64-bit atomicCAS / atomicAdd / plain RMW rate vs footprint on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint64_t mix64(uint64_t z){ z=(z^(z>>30))*0xBF58476D1CE4E5B9ULL; z=(z^(z>>27))*0x94D049BB133111EBULL; return z^(z>>31);} 
template<int MODE> __global__ void k(unsigned long long* t, uint64_t mask, uint64_t n, unsigned long long* sink){
  unsigned long long acc=0;
  for(uint64_t i=(uint64_t)blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=(uint64_t)gridDim.x*blockDim.x){
    uint64_t pos=mix64(i*0x9E3779B97F4A7C15ULL+12345)&mask;
    if(MODE==0) acc+=atomicCAS(&t[pos],0ULL,(unsigned long long)(i|1));
    else if(MODE==1) acc+=atomicAdd(&t[pos],1ULL);
    else if(MODE==2) atomicAdd(&t[pos],1ULL);           // no return
    else if(MODE==3) { unsigned long long v=t[pos]; t[pos]=v+i; }  // plain RMW (racy, rate only)
    else if(MODE==4) acc+=t[pos];                         // plain load
  }
  if(acc==0x1234567) *sink=acc;
}
int main(){
  unsigned long long *t,*sink; size_t maxb=(size_t)16<<30; hipMalloc(&t,maxb); hipMalloc(&sink,8);
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  const uint64_t n=(uint64_t)1<<29;
  const char* names[5]={"cas_ret","add_ret","add_noret","plain_rmw","plain_load"};
  for(int lg=14; lg<=31; lg+= (lg<20?3:(lg<28?4:1))){
    uint64_t slots=1ULL<<lg; 
    for(int mode=0;mode<5;++mode){
      hipMemset(t,0,slots*8); hipDeviceSynchronize();
      hipEventRecord(a);
      switch(mode){case 0:k<0><<<2048,256>>>(t,slots-1,n,sink);break;case 1:k<1><<<2048,256>>>(t,slots-1,n,sink);break;case 2:k<2><<<2048,256>>>(t,slots-1,n,sink);break;case 3:k<3><<<2048,256>>>(t,slots-1,n,sink);break;default:k<4><<<2048,256>>>(t,slots-1,n,sink);}
      hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b);
      printf("slots=2^%d (%8.1f MiB) %-10s %8.2f ms  %7.2f Gops/s\n",lg,slots*8/1048576.0,names[mode],ms,n/ms/1e6);
    }
  }
  return 0;
}
