// CPU check (gcc -O2 -march=native -ffp-contract=off markstein.c -lm): division by a per-level constant through its correctly rounded
// reciprocal + one fused residual step equals the IEEE quotient for the 16 voxel sizes of the NGP configuration (weights and cell
// coordinates), and the near-integer test of the gather kernel catches every case where floor(n * r) differs from floor(n / vs).
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
static inline uint64_t rng(uint64_t *s){ *s ^= *s<<13; *s ^= *s>>7; *s ^= *s<<17; return *s; }
int main(){
  int res[16]={15,22,30,42,58,80,111,153,212,294,406,561,776,1072,1482,2047};
  uint64_t s=88172645463325252ull; long bad1=0,bad2=0,tot=0;
  for(int l=0;l<16;l++){
    volatile float vs = 2.0f/(float)res[l]; volatile float r = 1.0f/vs;
    for(long i=0;i<60000000;i++){
      uint32_t u=(uint32_t)rng(&s);
      float frac=(u>>8)*(1.0f/16777216.0f);
      float num = frac*vs*1.001f; if(i&1) num = num*1e-3f; if((i&7)==3) num=-num;
      if((i&15)==5){ uint32_t b; float t=vs*(float)((u>>3)%3); memcpy(&b,&t,4); b += (u&7)-3; memcpy(&num,&b,4);}  // near multiples
      float q = num/vs;
      float q0 = num*r; float e=fmaf(-q0,vs,num); float q1=fmaf(e,r,q0);
      float e2=fmaf(-q1,vs,num); float q2=fmaf(e2,r,q1);
      if (isnan(q) || fabsf(num) < 1e-30f) continue; bad1 += (q1!=q); bad2 += (q2!=q); tot++;
    }
  }
  printf("tot %ld markstein1 mismatches %ld  two-step mismatches %ld\n",tot,bad1,bad2);
  // cell-index style: v = n/vs with n in [0,2]
  bad1=bad2=0;tot=0;
  for(int l=0;l<16;l++){
    volatile float vs = 2.0f/(float)res[l]; volatile float r = 1.0f/vs;
    for(long i=0;i<30000000;i++){
      uint32_t u=(uint32_t)rng(&s);
      float num=(u>>8)*(1.0f/16777216.0f)*2.0f;
      float q = num/vs;
      float q0 = num*r; float e=fmaf(-q0,vs,num); float q1=fmaf(e,r,q0);
      bad1 += (q1!=q); tot++;
      if (floorf(q0)!=floorf(q)) { float d=fabsf(q0-rintf(q0)); if(!(d <= 4e-7f*fmaxf(fabsf(q0),1.0f))) bad2++; }
    }
  }
  printf("cell: tot %ld markstein1 mismatches %ld  undetected floor mismatches %ld\n",tot,bad1,bad2);
}
