// asan_host.cpp -- the host half of the C ABI (nchmm_host.cpp: no HIP) under AddressSanitizer + UBSan, edge sizes included.
// GPU sanitizers are not available on the pool, so this is the sanitized build the device-free code gets:
//   make -C tools asan-host   (tests/test_host_prep.py::test_host_abi_under_sanitizers runs it)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>
#include "nanocall_hip.h"
int main(){
  std::mt19937 rng(1); std::uniform_real_distribution<float> u(0.f,1.f);
  std::vector<float> table(4096*4); for(unsigned j=0;j<4096;++j){table[4*j]=45+50*u(rng);table[4*j+1]=1+u(rng);table[4*j+2]=.9f+.6f*u(rng);table[4*j+3]=.3f+.2f*u(rng);}
  std::vector<float> st(4096*10), t6(4096*6); float pm[6]={1.02f,-.7f,.001f,1.05f,.95f,1.3f};
  if(nchmm_model_load(table.data(),st.data())) return 1; if(nchmm_model_scale(st.data(),pm)) return 2; if(nchmm_model_pack6(st.data(),t6.data())) return 3;
  std::vector<uint32_t> rp(4097); std::vector<uint16_t> pred(NCHMM_MAX_ARCS); std::vector<float> w(NCHMM_MAX_ARCS); uint32_t na=0;
  if(nchmm_transitions_fast(.3f,.1f,rp.data(),pred.data(),w.data(),&na)) return 4; printf("arcs %u\n",na);
  for(size_t n: {0ul,1ul,7ul,70000ul,200001ul}){ std::vector<float> m(n),s(n),t(n),cm(n),ls(n); for(size_t i=0;i<n;++i){m[i]=60+u(rng);s[i]=(i%97==0)?0.f:1+u(rng);t[i]=i*.01f;}
    if(nchmm_events_prepare(n,m.data(),s.data(),t.data(),.002f,cm.data(),ls.data())) return 5; if(nchmm_events_prepare(n,m.data(),s.data(),nullptr,0.f,cm.data(),ls.data())) return 6; }
  for(size_t n: {0ul,1ul,2ul,5000ul}){ std::vector<uint16_t> states(n); unsigned k=7; for(size_t i=0;i<n;++i){ float r=u(rng); if(r>=.1f) k=r<.7f? ((k<<2)|(rng()&3))&4095 : ((k<<4)|(rng()&15))&4095; states[i]=k;}
    std::vector<int32_t> mv(n); std::vector<char> seq(6*n+8); size_t len=0; if(nchmm_base_seq(n,states.data(),mv.data(),seq.data(),&len)) return 7;
    std::vector<char> out(len+len/80+64); size_t on=0; if(n && nchmm_write_fasta("r:f:0",seq.data(),80,out.data(),out.size(),&on)) return 8;
    size_t need=0; int rc=nchmm_write_fasta("r:f:0",seq.data(),80,out.data(),3,&need); (void)rc; }
  uint16_t km[4096]; uint32_t nk=0; if(nchmm_st_train_kmers(km,&nk)) return 9; printf("train kmers %u\n",nk);
  { size_t n=400; std::vector<float> sums(6*n),mean(n),sd(n),start(n); for(size_t i=0;i<n;++i){ for(int q=0;q<6;++q) sums[6*i+q]=1+u(rng); mean[i]=60+u(rng); sd[i]=1+u(rng); start[i]=i*.01f;}
    float crt[6]={1,0,0,1,1,1}, np_[6]; int done=0; if(nchmm_train_pm_finish(n,sums.data(),mean.data(),sd.data(),start.data(),1,crt,np_,&done)) return 10;
    if(nchmm_train_pm_finish(n,sums.data(),mean.data(),sd.data(),nullptr,0,crt,np_,&done)) return 11;
    std::vector<float> z(6*n,0.f); if(nchmm_train_pm_finish(n,z.data(),mean.data(),sd.data(),start.data(),1,crt,np_,&done)) return 12; printf("singular done=%d\n",done);
    float st3[12]={-1,-2,-3,-1,-2,-3,-1,-2,-3,-1,-2,-3}, ps,pk; if(nchmm_train_st_finish(4,st3,&ps,&pk)) return 13; if(nchmm_train_st_finish(0,nullptr,&ps,&pk)) return 14; }
  puts("host ABI under ASan/UBSan: ok"); return 0; }
