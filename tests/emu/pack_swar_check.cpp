// pack_swar_check.cpp - TEST INFRASTRUCTURE: metagraph_amd/csrc/pack_swar.hpp (k_pack_reads) against the byte loop of graph_build.hpp pack_read_word,
// restated here with KmerExtractorBOSS::encode, on random and adversarial bytes.  Built and run by tests/test_pack_swar.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include "../../metagraph_amd/csrc/pack_swar.hpp"
static uint32_t enc(uint8_t ch){ if (ch&0x80) return 5; switch(ch){case 'A':case 'a':return 1;case 'C':case 'c':return 2;case 'G':case 'g':return 3;case 'T':case 't':case 'U':case 'u':return 4;default:return 5;} }
static uint32_t scode(const char*seq,int L,int strand,int pos){ if(!strand) return enc((uint8_t)seq[pos]); uint32_t c=enc((uint8_t)seq[L-1-pos]); return c==5?5u:5u-c; }
static void ref(const char*seq,int L,int strand,int j,uint64_t*codes,uint32_t*inv){ uint64_t c=0; uint32_t v=0; for(int t=0;t<32;++t){int pos=32*j+t; if(pos>=L)break; uint32_t code=scode(seq,L,strand,pos); if(code==5||code==0) v|=1u<<t; else c|=(uint64_t)(code-1)<<(2*t);} *codes=c;*inv=v; }
int main(){ srand(7); const char al[]="ACGTacgtUuNnXx-*\x01\xc1\x41\x61\x7f@[`{"; long n=0;
  for(int it=0;it<400000;++it){ int L=1+rand()%200; char buf[300]; memset(buf,'G',sizeof buf); char*seq=buf+40; for(int i=-40;i<L+40;++i){ int r=rand()%100; seq[i]= r<80? "ACGT"[rand()%4] : (r<95? al[rand()%(sizeof(al)-1)] : (char)(rand()%256)); }
    int nw=(L+31)/32; for(int j=0;j<nw;++j) for(int s=0;s<2;++s){ uint64_t c0; uint32_t v0; ref(seq,L,s,j,&c0,&v0); int nf=L-32*j<32?L-32*j:32; uint64_t b[4]; const char*p = s? seq+(L-32*j-32) : seq+32*j; memcpy(b,p,32); uint64_t c1; uint32_t v1; mgx_pack::pack32(b,nf,s,&c1,&v1); ++n; if(c0!=c1||v0!=v1){ printf("MISMATCH L %d j %d s %d: %016llx %08x vs %016llx %08x\n",L,j,s,(unsigned long long)c0,v0,(unsigned long long)c1,v1); return 1; } } }
  printf("ok %ld words\n",n); return 0; }
