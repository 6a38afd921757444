// scratch: per-read statistics of the oracle on a config-1-like workload
#include "../oracle/xmo_worker.h"
#include <cstdio>
#include <random>
using namespace xmo;
static uint64_t sm(uint64_t& s){ s += 0x9E3779B97F4A7C15ull; uint64_t z=s; z=(z^(z>>30))*0xBF58476D1CE4E5B9ull; z=(z^(z>>27))*0x94D049BB133111EBull; return z^(z>>31);}
int main(int argc,char**argv){
  int refLen = argc>1?atoi(argv[1]):5000000; int nReads = argc>2?atoi(argv[2]):2000; int readLen= argc>3?atoi(argv[3]):150;
  uint64_t s=0xEC011; std::string text(refLen,'A'); const char* B="ACGT"; for(int i=0;i<refLen;i++) text[i]=B[sm(s)>>62];
  ReferenceDatabase ref; ref.sequences.addForward(makeSequence("ecoli_syn", text)); ref.finish(false,true);
  AlignmentParameters p; p.MutationPenalty=1; p.InsertionStart_Penalty=1.5; p.InsertionExtension_Penalty=0.6; p.DeletionStart_Penalty=1.5; p.DeletionExtension_Penalty=0.5; p.MaxErrorRate=0.1; p.UnalignedPenalty=0.1; p.AmbiguityPenalty=0.1; p.Max_PenaltySpan=0.5; p.MaxNumMatches=INT32_MAX;
  AlignerWorker w(&ref,p);
  fprintf(stderr,"index ready minInteresting=%d maxHashed=%d\n", ref.hashblockDatabase->minInterestingSize, ref.hashblockDatabase->maxFullySetUpSize);
  uint64_t rs=0x5EED0001;
  long hist_rows[64]={0}; long totBlocks=0, totProbes=0, totFetch=0, totHits=0, totCand=0, totPA=0, totNodes=0, quick=0; long hProbe[64]={0}; long maxRows=0; long steps=0;
  long lvlBlocks[64]={0};
  for(int r=0;r<nReads;r++){
    int start = sm(rs)%(refLen-readLen-4); bool rev = sm(rs)&1;
    std::string t = text.substr(start, readLen+4);
    std::string rd;
    bool indel = (sm(rs)%100)<5; int ipos = 10+sm(rs)%(readLen-20); int ilen=1+sm(rs)%3; bool ins = sm(rs)&1;
    for(int i=0;(int)rd.size()<readLen;i++){ if(indel && i==ipos){ if(ins){ for(int k=0;k<ilen;k++) rd.push_back(B[sm(rs)>>62]); } else { i+=ilen; } } char c=t[i]; if(sm(rs)%100<1){ c=B[(strchr(B,c)-B+1+sm(rs)%3)&3]; } rd.push_back(c);} rd.resize(readLen);
    std::vector<uint8_t> codes(readLen); for(int i=0;i<readLen;i++) codes[i]=Basepairs::encode(rd[i]);
    if(rev){ std::vector<uint8_t> c2(readLen); for(int i=0;i<readLen;i++) c2[i]=Basepairs::complement(codes[readLen-1-i]); codes=c2; }
    AlignerWorker::QueryContext ctx; ctx.mates.emplace_back(new QuerySequence("q",codes)); ctx.query.sequences.push_back(ctx.mates.back()->fwd.get());
    Counters before=w.counters;
    QueryAlignments qa = w.alignToAncestralReference(ctx);
    Counters& c=w.counters;
    long probes=c.headerProbes-before.headerProbes; totProbes+=probes; hProbe[std::min(63L,probes/4)]++;
    totFetch+=c.bucketFetches-before.bucketFetches; totHits+=c.hitsFetched-before.hitsFetched; totCand+=c.candidatesExtended-before.candidatesExtended; totPA+=c.pathAlignerCalls-before.pathAlignerCalls; totNodes+=c.pathAlignerNodes-before.pathAlignerNodes; quick+=c.quickAccepts-before.quickAccepts;
    auto& pyr = ctx.components[0]->pyramid; long rows=pyr.rows.size(); hist_rows[std::min(63L,rows)]++; if(rows>maxRows) maxRows=rows;
    for(size_t k=1;k<pyr.rows.size();k++){ auto* pr = dynamic_cast<HashBlock_ParentRow*>(pyr.rows[k].get()); if(pr){ totBlocks+=pr->blockList.size(); lvlBlocks[std::min((size_t)63,k)]+=pr->blockList.size(); } }
    steps += ctx.components[0]->interestingMatch_history.size();
  }
  printf("reads %d: probes/read %.2f fetch %.2f hits %.2f cand %.2f PAcalls %.3f nodes/read %.1f quick %.3f steps %.2f lazyParentBlocks/read %.1f maxRows %ld\n", nReads,(double)totProbes/nReads,(double)totFetch/nReads,(double)totHits/nReads,(double)totCand/nReads,(double)totPA/nReads,(double)totNodes/nReads,(double)quick/nReads,(double)steps/nReads,(double)totBlocks/nReads,maxRows);
  printf("rows hist:"); for(int i=0;i<64;i++) if(hist_rows[i]) printf(" %d:%ld",i,hist_rows[i]); printf("\n");
  printf("probe hist (x4):"); for(int i=0;i<64;i++) if(hProbe[i]) printf(" %d:%ld",i*4,hProbe[i]); printf("\n");
  printf("blocks per level per read:"); for(int i=0;i<64;i++) if(lvlBlocks[i]) printf(" L%d:%.1f",i,(double)lvlBlocks[i]/nReads); printf("\n");
}
