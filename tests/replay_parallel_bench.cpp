// Micro-benchmark of the host replay, serial against the stages of ParallelReplay (csrc/adsb_replay_host.h), one thread, over the
// records a dense capture leaves (cut out of the oracle's orc_all_trials by tests/replay_parallel_bench.sh).  Not a test.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>
#include "../dump1090_rs_amd/csrc/adsb_replay_host.h"
using namespace adsb; using namespace adsb::host;
static double now(){return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();}
int main(int argc,char**argv){
  int copies = argc>1?atoi(argv[1]):64;
  FILE*f=fopen("/tmp/recs.bin","rb"); std::vector<TrialRecord> all; TrialRecord r; while(fread(&r,32,1,f)==1) all.push_back(r); fclose(f);
  Crc24 crc; std::set<uint32_t> addrs; std::vector<TrialRecord> keep;
  auto resid=[&](const TrialRecord&t){return crc.residual(t.msg,(t.msg[0]&0x80)?14:7);};
  for(auto&t:all){uint32_t df=t.msg[0]>>3; if((df==17||df==18||df==11)&&resid(t)==0) addrs.insert(uint32_t(t.msg[1])<<16|uint32_t(t.msg[2])<<8|t.msg[3]);}
  for(auto&t:all){uint32_t df=t.msg[0]>>3; uint32_t c=resid(t); bool sv=(df==17||df==18)&&c==0; bool d11=df==11&&(c&0xFFFF80)==0; bool ap=((0xFF310031u>>df)&1)&&addrs.count(c);
    if(sv||d11||ap){ t.power=(t.power&((1ull<<40)-1))|((uint64_t)c<<40); t.pad=1; keep.push_back(t);} }
  std::vector<TrialRecord> sorted; if(sort_records(keep.data(),keep.size(),sorted)) keep=sorted;
  std::vector<RecordRun> runs; for(int k=0;k<copies;k++) runs.push_back({keep.data(),keep.size(),(uint64_t)k*64});
  size_t total=keep.size()*copies;
  IcaoFilter f2; ParallelReplay pr; pr.plan(f2,crc,runs,28,true);
  for(int rep=0;rep<5;rep++){
    double s0=now(); for(int i=0;i<pr.parts();i++) pr.scan_part(i); double s1=now(); pr.merge(); double s2=now(); for(int i=0;i<pr.parts();i++) pr.score_part(i); double s3=now();
    IcaoFilter f1; std::vector<adsb_msg> o1; o1.reserve(total); double t0=now(); for(auto&rr:runs) replay_sorted(f1,crc,rr.rec,rr.n,rr.chunk_offset,o1); double t1=now();
    printf("records %zu: scan %.2f ns/rec, merge %.0f us, score %.2f ns/rec, serial %.2f ns/rec\n", total,(s1-s0)*1e9/total,(s2-s1)*1e6,(s3-s2)*1e9/total,(t1-t0)*1e9/total);
  }
}
