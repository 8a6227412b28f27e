// asan_host.cpp -- the host half of the C ABI (nchmm_host.cpp, nchmm_reads.cpp: no HIP) under AddressSanitizer + UBSan, edge sizes included.
// GPU sanitizers are not available on the pool, so this is the sanitized build the device-free code gets:
//   make -C tools asan-host   (tests/test_host_prep.py::test_host_abi_under_sanitizers runs it)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <atomic>
#include <thread>
#include <chrono>
#include <vector>
#include "nchmm_internal.hpp"
#include "nchmm_combine.hpp"
#include "nchmm_plan.hpp"
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
  // read summary (nchmm_reads.cpp): empty / tiny / ordinary 2D / no hairpin / all-abasic tables, both presets, --1d, trims larger than the read
  for (const char* pore : {"r73", "r9"}) for (unsigned one_d : {0u, 1u}) for (size_t n : {0ul, 1ul, 9ul, 60ul, 400ul, 3000ul}) for (int shape = 0; shape < 4; ++shape) {
    nchmm_segment_opts so; if (nchmm_segment_opts_default(&so, pore)) return 20; so.template_only = one_d;
    if (shape == 3) { so.trim_margins[0] = so.trim_margins[1] = 5000; so.max_ed_events = 100; }
    std::vector<nchmm_ed_event> ed(n); int64_t at = 0;
    for (size_t i = 0; i < n; ++i) {
      const bool hp = shape == 0 && n >= 400 && i >= n / 2 && i < n / 2 + 12;   // a hairpin island of abasic-level events in the middle
      const double lvl = shape == 2 ? 130.0 : (hp ? 125.0 + u(rng) : 55.0 + 30.0 * u(rng));
      ed[i].mean = lvl; ed[i].stdv = (i % 53 == 0) ? 0.0 : 0.8 + u(rng); ed[i].start = at; ed[i].length = 3 + (int64_t)(rng() % 40); at += ed[i].length; }
    nchmm_read_summary rs; if (nchmm_read_summarize(&so, n, ed.data(), 4000.f, 1, &rs)) return 21;
    if (rs.num_ed_events > n) return 22;
    for (int st = 0; st < 2; ++st) {
      const size_t cap = rs.strand_bounds[2 * st + 1] >= rs.strand_bounds[2 * st] ? rs.strand_bounds[2 * st + 1] - rs.strand_bounds[2 * st] : 0;
      std::vector<float> m(cap + 1), sd(cap + 1), stt(cap + 1), len(cap + 1); size_t got = 0;
      if (rs.num_ed_events && nchmm_read_load_events(&rs, ed.data(), 4000.f, st, m.data(), sd.data(), stt.data(), len.data(), &got)) return 23;
      if (got > cap) return 24;
      float mean = 0, stdv = 0; if (nchmm_mean_stdv(got, m.data(), &mean, &stdv)) return 25; }
  }
  { float sc, sh; const float r0[2] = {70.f, 9.f}, r1[2] = {68.f, 10.f}, m0[2] = {66.f, 11.f}, m1[2] = {65.f, 12.f}, z[2] = {0.f, 0.f};
    if (nchmm_initial_scaling(1, r0, r1, m0, m1, &sc, &sh)) return 26; if (nchmm_initial_scaling(0, r0, nullptr, m0, nullptr, &sc, &sh)) return 27;
    (void)nchmm_initial_scaling(0, z, nullptr, z, nullptr, &sc, &sh);   // degenerate: must not trap
    float mean, stdv; if (nchmm_mean_stdv(0, nullptr, &mean, &stdv)) return 28; if (nchmm_mean_stdv(1, r0, &mean, &stdv)) return 29; }
  { // the host worker pool with several callers at once (one host thread per device runs such loops, nchmm_pool.cpp):
    // every index of every loop is visited exactly once, slots are reused thousands of times
    std::atomic<long> bad{0};
    std::vector<std::thread> callers;
    for (int id = 0; id < 6; ++id)
      callers.emplace_back([&bad, id] {
        for (int rep = 0; rep < 400; ++rep) {
          const size_t n = 5 + (size_t)((rep * 7 + id * 13) % 600);
          std::vector<int> hit(n, 0);
          nchmm::parallel_for(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) hit[i] += 1; });
          for (size_t i = 0; i < n; ++i) if (hit[i] != 1) bad++;
        } });
    for (auto& t : callers) t.join();
    if (bad.load()) return 30; }
  { // StrandCombiner (nchmm_viterbi_strand's batcher) with a host stand-in for the device: 24 threads x 150 strands of random
    // length, small batches (so that they fill up, close early, and a strand longer than a batch comes along); every caller
    // must get ITS strand's results -- states derived from its own events, logp from its own image and transition parameters
    struct FakeRunner {
      std::atomic<long>* runs; std::atomic<long>* strands;
      void release(nchmm::StrandBatch& B) { std::free(B.images); std::free(B.fast); std::free(B.p_skip); std::free(B.p_stay); std::free(B.off); std::free(B.cm); std::free(B.sd); std::free(B.ls);
        std::free(B.states); std::free(B.logp); std::free(B.status); std::free(B.base); std::free(B.scale6); B.base = nullptr; B.scale6 = nullptr;
        B.images = nullptr; B.fast = nullptr; B.p_skip = B.p_stay = nullptr; B.off = nullptr; B.cm = B.sd = B.ls = nullptr; B.states = nullptr; B.logp = nullptr; B.status = nullptr;
        B.cap[0] = B.cap[1] = B.cap[2] = 0; }
      int alloc(nchmm::StrandBatch& B, const size_t cap[3]) { release(B); const size_t reads = cap[0], events = cap[1];
        B.images = (float*)std::malloc(sizeof(float) * nchmm::kImageFloats * cap[2]); B.fast = (int32_t*)std::malloc(4 * cap[2]); B.p_skip = (float*)std::malloc(4 * reads);
        B.p_stay = (float*)std::malloc(4 * reads); B.off = (uint64_t*)std::malloc(8 * (reads + 1)); B.cm = (float*)std::malloc(4 * events); B.sd = (float*)std::malloc(4 * events);
        B.ls = (float*)std::malloc(4 * events); B.states = (uint16_t*)std::malloc(2 * events); B.logp = (float*)std::malloc(4 * reads); B.status = (int32_t*)std::malloc(4 * reads); B.base = (const float**)std::malloc(8 * reads); B.scale6 = (float*)std::malloc(24 * reads);
        B.off[0] = 0; B.cap[0] = reads; B.cap[1] = events; B.cap[2] = cap[2]; return 0; }
      int run(nchmm::StrandBatch& B) { runs->fetch_add(1); const size_t n = B.used[0]; strands->fetch_add((long)n);
        for (size_t r = 0; r < n; ++r) { for (uint64_t e = B.off[r]; e < B.off[r + 1]; ++e) B.states[e] = (uint16_t)((unsigned)B.cm[e] & 4095u);
          B.logp[r] = B.images[r * nchmm::kImageFloats + 17] + B.p_skip[r] * 8.f + B.p_stay[r]; B.status[r] = (B.off[r + 1] - B.off[r]) % 7 == 3 ? -6 : 0; }
        std::this_thread::sleep_for(std::chrono::microseconds(300)); return 0; } };
    std::atomic<long> runs{0}, strands{0}, bad{0};
    {
      nchmm::StrandCombiner<FakeRunner> sc(FakeRunner{&runs, &strands}, 8, 600, 50);
      std::vector<std::thread> callers;
      for (int id = 0; id < 24; ++id)
        callers.emplace_back([&sc, &bad, id] {
          std::mt19937 r(100 + id);
          for (int rep = 0; rep < 150; ++rep) {
            const size_t n = 1 + r() % ((rep % 40 == 7) ? 1500 : 200);      // (some longer than a whole batch)
            std::vector<float> cm(n), sd(n, 1.f), ls(n, 0.f); for (size_t i = 0; i < n; ++i) cm[i] = (float)((id * 131 + rep * 17 + i) & 4095);
            std::vector<uint16_t> st(n, 0xFFFF); float lp = -1; const float tag = (float)(id * 1000 + rep);
            const int rc = sc.submit([&](float* img, int32_t* fast) { img[17] = tag; *fast = 1; }, (float)(id % 5), (float)(rep % 3), n, cm.data(), sd.data(), ls.data(), st.data(), &lp);
            if (rc != (n % 7 == 3 ? -6 : 0)) bad++;
            if (lp != tag + (float)(id % 5) * 8.f + (float)(rep % 3)) bad++;
            for (size_t i = 0; i < n; ++i) if (st[i] != (uint16_t)((id * 131 + rep * 17 + i) & 4095)) { bad++; break; }
          } });
      for (auto& t : callers) t.join();
    }
    printf("combiner: %ld strands in %ld batches\n", strands.load(), runs.load());
    if (bad.load() || strands.load() != 24 * 150 || runs.load() >= strands.load()) return 31; }
  { // the batch plan (nchmm_plan.hpp) on every shape: ranges partition the reads, the order is a permutation that is longest-first
    // inside each range, outliers are exactly the reads longer than a pooled region may hold, a prefix of every range's order
    std::mt19937 r(99);
    for (int trial = 0; trial < 400; ++trial) {
      const size_t n = 1 + r() % (trial % 7 == 0 ? 5000 : 300), slots = 1 + r() % 600;
      std::vector<uint64_t> off(n + 1, 0);
      for (size_t i = 0; i < n; ++i) { uint64_t len = r() % 5 == 0 ? 0 : 1 + r() % 4000; if (r() % 97 == 0) len = 20000 + r() % 200000; off[i + 1] = off[i] + len; }
      for (int alone = 0; alone < 2; ++alone) for (size_t forced : {(size_t)0, (size_t)1, (size_t)7}) {
        std::vector<nchmm::PipeRange> rg; nchmm::cut_ranges(off.data(), n, slots, alone != 0, forced, &rg);
        if (rg.empty() || rg.front().r0 != 0 || rg.back().r1 != n) return 40;
        for (size_t k = 0; k < rg.size(); ++k) { if (rg[k].r1 <= rg[k].r0 || (k && rg[k].r0 != rg[k - 1].r1) || rg[k].e0 != off[rg[k].r0] || rg[k].e1 != off[rg[k].r1]) return 41;
          size_t mx = 0; for (size_t i = rg[k].r0; i < rg[k].r1; ++i) mx = std::max<size_t>(mx, (size_t)(off[i + 1] - off[i])); if (mx != rg[k].max_events) return 42; }
        if (!forced && rg.size() > 2) return 43;
        std::vector<uint32_t> order; nchmm::order_ranges(off.data(), n, rg, &order);
        std::vector<char> seen(n, 0); for (uint32_t v : order) { if (v >= n || seen[v]) return 44; seen[v] = 1; }
        for (const auto& g : rg) for (size_t i = g.r0; i < g.r1; ++i) { if (order[i] < g.r0 || order[i] >= g.r1) return 45;
          if (i > g.r0 && off[order[i] + 1] - off[order[i]] > off[order[i - 1] + 1] - off[order[i - 1]]) return 46; }
        const size_t pool = 8 * (1 + r() % 72), row = 4096, budget = ((size_t)1 << 20) * (16 + r() % 200000);
        nchmm::OutlierPlan P = nchmm::plan_outliers(off.data(), n, rg, order, pool, budget, row);
        uint64_t longest = 1; for (size_t i = 0; i < n; ++i) longest = std::max<uint64_t>(longest, off[i + 1] - off[i]);
        if (P.n_out.size() != rg.size()) return 47;
        size_t tot = 0; for (size_t k = 0; k < rg.size(); ++k) tot += P.n_out[k];
        if (tot != P.outliers.size() || tot * 8 > n) return 48;
        if (P.outliers.empty()) { if (P.pool_longest != longest) return 49; }
        else {
          const uint64_t cap = (uint64_t)(budget / 10 * 7 / pool / row);
          if (P.pool_longest > cap || P.pool_longest * row > budget / pool || P.budget_big == 0) return 50;
          std::vector<char> is_out(n, 0); for (uint32_t v : P.outliers) { if (off[v + 1] - off[v] <= cap || is_out[v]) return 51; is_out[v] = 1; }
          for (size_t i = 0; i < n; ++i) if (!is_out[i] && off[i + 1] - off[i] > cap) return 52;
          for (size_t k = 0; k < rg.size(); ++k) for (size_t i = 0; i < rg[k].r1 - rg[k].r0; ++i) if ((i < P.n_out[k]) != (is_out[order[rg[k].r0 + i]] != 0)) return 53;
          for (size_t i = 1; i < P.outliers.size(); ++i) if (off[P.outliers[i] + 1] - off[P.outliers[i]] > off[P.outliers[i - 1] + 1] - off[P.outliers[i - 1]]) return 54;
        }
      }
    }
    puts("batch plans: ok"); }
  { // which form of the sweep a launch takes (nchmm_plan.hpp: lpt_makespan_us, choose_sweep, choose_sweep_bounds)
    std::mt19937 r(7);
    const nchmm::SweepRates R;
    for (int trial = 0; trial < 300; ++trial) {
      const size_t n = 1 + r() % (trial % 9 == 0 ? 20000 : 1500), slots = 1 + r() % 600;
      std::vector<uint64_t> lens(n);
      uint64_t longest = 0, total = 0;
      for (auto& l : lens) { l = r() % 7 == 0 ? 0 : 1 + r() % 6000; if (r() % 211 == 0) l = 30000 + r() % 100000; longest = std::max(longest, l); total += l; }
      // a schedule is never shorter than its longest read nor than its share of the work, and never longer than both together
      const double us = 0.5 + (r() % 100) / 50.0, per_read = r() % 2 ? 0.0 : 40.0;
      const double t = nchmm::lpt_makespan_us(lens, slots, us, per_read);
      const double lo = std::max((double)longest * us + per_read, ((double)total * us + (double)n * per_read) / (double)std::max(slots, std::min(n, slots)));
      if (t < lo * (1 - 1e-9) - 1e-6 && n > slots) return 60;
      if (t > (double)longest * us + per_read + ((double)total * us + (double)n * per_read) / (double)slots + 1e-6) return 61;
      if (nchmm::lpt_makespan_us(lens, slots + 1, us, per_read) > t * (1 + 1e-9) + 1e-6 && n <= 16384) return 62;   // more blocks never hurt
      // the decision: a launch that fits one read per CU takes the low-latency form; equal reads that fill the wide sweep's
      // slots twice over take the wide form; a streaming caller's full launches stay wide whatever their shape
      const size_t n_cu = 256, wide = 512;
      const nchmm::Sweep c = nchmm::choose_sweep(lens, n_cu, wide, false, R);
      if (c != nchmm::kSweepWide && c != nchmm::kSweepLl) return 63;
      if (n <= n_cu && longest > 1000 && c != nchmm::kSweepLl) return 64;
      if (n * 3 >= wide && nchmm::choose_sweep(lens, n_cu, wide, true, R) != nchmm::kSweepWide) return 65;
      std::vector<uint64_t> equal(4 * wide, 5000);
      if (nchmm::choose_sweep(equal, n_cu, wide, false, R) != nchmm::kSweepWide) return 66;
      equal.resize(n_cu);
      if (nchmm::choose_sweep(equal, n_cu, wide, false, R) != nchmm::kSweepLl || nchmm::choose_sweep_bounds(n_cu, 5000, n_cu * 5000, n_cu, wide, false, R) != nchmm::kSweepLl) return 67;
      if (nchmm::choose_sweep_bounds(4 * wide, 5000, 4 * wide * 5000, n_cu, wide, false, R) != nchmm::kSweepWide) return 68;
      // one read that is longer than everything else together sets the duration either way: the form that halves it wins
      std::vector<uint64_t> skew(1000, 2000); skew[17] = 400000;
      if (nchmm::choose_sweep(skew, n_cu, wide, false, R) != nchmm::kSweepLl || nchmm::choose_sweep_bounds(1000, 400000, 1000 * 2000 + 398000, n_cu, wide, false, R) != nchmm::kSweepLl) return 69;
      if (nchmm::choose_sweep(std::vector<uint64_t>(), n_cu, wide, false, R) != nchmm::kSweepWide || nchmm::choose_sweep_bounds(0, 0, 0, n_cu, wide, false, R) != nchmm::kSweepWide) return 70;
    }
    // emissions ahead (plan_ahead): never more reads than the buffer holds or than kMaxAheadReads, never when it does not pay
    for (int trial = 0; trial < 200; ++trial) {
      const size_t n = 1 + r() % 700;
      std::vector<uint64_t> lens(n);
      for (auto& l : lens) l = 1 + r() % (trial % 3 ? 3000 : 40000);
      std::sort(lens.begin(), lens.end(), std::greater<uint64_t>());
      const uint64_t budget = (r() % 4 == 0) ? 0 : 1 + r() % 60000;
      double t = -1;
      const size_t K = nchmm::plan_ahead(lens, 256, budget, &t, R);
      uint64_t rows = 0; for (size_t k = 0; k < K; ++k) rows += lens[k];
      if (K > n || K > nchmm::kMaxAheadReads || rows > budget || t <= 0) return 71;
      const double t0 = nchmm::lpt_makespan_us(lens, 256, R.ll, R.per_read_us);
      if (K == 0 ? t != t0 : t >= t0) return 72;
      size_t k2 = 99;
      const nchmm::Sweep c = nchmm::choose_sweep(lens, 256, 512, false, R, budget, &k2);
      if ((c == nchmm::kSweepAhead) != (k2 > 0) || (c == nchmm::kSweepAhead && k2 != K)) return 73;
      if (nchmm::choose_sweep(lens, 256, 512, false, R, 0, &k2) == nchmm::kSweepAhead || k2 != 0) return 74;
    }
    { // one strand: ahead when it fits the buffer
      std::vector<uint64_t> one(1, 5000);
      size_t k = 0;
      if (nchmm::choose_sweep(one, 256, 512, false, R, 16384, &k) != nchmm::kSweepAhead || k != 1) return 75;
      if (nchmm::choose_sweep_bounds(1, 5000, 5000, 256, 512, false, R, 16384) != nchmm::kSweepAhead) return 76;
      if (nchmm::choose_sweep_bounds(1, 50000, 50000, 256, 512, false, R, 16384) != nchmm::kSweepLl) return 77;
      if (nchmm::choose_sweep_bounds(1, 5000, 5000, 256, 512, true, R, 16384) == nchmm::kSweepAhead) return 78;
    }
    puts("sweep choice: ok"); }
  puts("host ABI under ASan/UBSan: ok"); return 0; }
