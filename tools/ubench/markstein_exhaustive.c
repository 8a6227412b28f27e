// exhaustive proof by enumeration: for EVERY binary32 divisor mantissa (high to low) and EVERY numerator mantissa,
// one_step(n, d) == n / d ?   (exponents/signs do not matter away from under/overflow)
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <omp.h>
static inline float one_step(float n, float d, float r){ float q=n*r; float e=fmaf(-q,d,n); return fmaf(e,r,q);} 
int main(int argc,char**argv){
  uint32_t hi = argc>1? strtoul(argv[1],0,0) : 0x7FFFFF, lo = argc>2? strtoul(argv[2],0,0) : 0;
  const uint32_t step = 4096;
  for (uint32_t top = hi; ; ) {
    uint32_t bot = top >= lo + step - 1 ? top - (step - 1) : lo;
    long fails = 0;
    #pragma omp parallel for schedule(dynamic,8) reduction(+:fails)
    for (uint32_t m = bot; m <= top; ++m) {
      uint32_t bits = (127u<<23) | m; float d; memcpy(&d,&bits,4);
      float r = (float)(1.0/(double)d);
      long f = 0;
      for (uint32_t nm=0; nm<(1u<<23); ++nm){
        uint32_t nb = (127u<<23)|nm; float n; memcpy(&n,&nb,4);
        float ref = n/d, got = one_step(n,d,r);
        uint32_t a,b; memcpy(&a,&ref,4); memcpy(&b,&got,4);
        f += (a != b);
      }
      if (f) { printf("FAIL divisor mant 0x%06x: %ld numerators\n", m, f); fflush(stdout); }
      fails += f;
    }
    printf("done divisors 0x%06x..0x%06x fails %ld\n", bot, top, fails); fflush(stdout);
    if (bot == lo) break;
    top = bot - 1;
  }
  return 0; }
