#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include "treeopt.hpp"
using namespace rt;
struct Leaf { float lo[3], hi[3], c[3]; };
static std::vector<Leaf> L;
static std::vector<OptNode> N;
static float area(const float* lo, const float* hi){ float x=hi[0]-lo[0],y=hi[1]-lo[1],z=hi[2]-lo[2]; return 2*(x*y+x*z+y*z);}
static int build(std::vector<uint32_t>& idx, uint32_t b, uint32_t e){
    int me = (int)N.size(); N.push_back(OptNode{});
    if (e-b==1){ OptNode& n=N[me]; for(int a=0;a<3;a++) n.lo[a]=L[idx[b]].lo[a], n.hi[a]=L[idx[b]].hi[a]; n.ref=idx[b]; n.height=1; return me; }
    uint32_t n=e-b; float best=INFINITY; int bax=-1; uint32_t bs=0;
    std::vector<float> ra(n);
    for(int ax=0;ax<3;ax++){
        std::sort(idx.begin()+b, idx.begin()+e, [&](uint32_t x,uint32_t y){return L[x].c[ax]<L[y].c[ax]||(L[x].c[ax]==L[y].c[ax]&&x<y);});
        float lo[3],hi[3];
        for(uint32_t i=n;i-->0;){ const Leaf&q=L[idx[b+i]]; if(i==n-1) for(int a=0;a<3;a++) lo[a]=q.lo[a],hi[a]=q.hi[a]; else for(int a=0;a<3;a++) lo[a]=std::min(lo[a],q.lo[a]),hi[a]=std::max(hi[a],q.hi[a]); ra[i]=area(lo,hi);}
        for(uint32_t i=0;i+1<n;i++){ const Leaf&q=L[idx[b+i]]; if(i==0) for(int a=0;a<3;a++) lo[a]=q.lo[a],hi[a]=q.hi[a]; else for(int a=0;a<3;a++) lo[a]=std::min(lo[a],q.lo[a]),hi[a]=std::max(hi[a],q.hi[a]);
            float c=area(lo,hi)*(i+1)+ra[i+1]*(n-i-1); if(c<best) best=c,bax=ax,bs=i+1; }
    }
    std::sort(idx.begin()+b, idx.begin()+e, [&](uint32_t x,uint32_t y){return L[x].c[bax]<L[y].c[bax]||(L[x].c[bax]==L[y].c[bax]&&x<y);});
    int l=build(idx,b,b+bs), r=build(idx,b+bs,e);
    N[me].child[0]=l; N[me].child[1]=r; N[l].parent=me; N[r].parent=me;
    for(int a=0;a<3;a++) N[me].lo[a]=std::min(N[l].lo[a],N[r].lo[a]), N[me].hi[a]=std::max(N[l].hi[a],N[r].hi[a]);
    N[me].height=1+std::max(N[l].height,N[r].height);
    return me;
}
int main(int argc,char**argv){
    FILE*f=fopen(argv[1],"rb"); fseek(f,0,SEEK_END); size_t n=ftell(f)/24; fseek(f,0,SEEK_SET);
    std::vector<float> raw(n*6); if(fread(raw.data(),4,n*6,f)!=n*6) return 1; fclose(f);
    L.resize(n); for(size_t i=0;i<n;i++){ for(int a=0;a<3;a++){ L[i].lo[a]=raw[6*i+a]; L[i].hi[a]=raw[6*i+3+a]; L[i].c[a]=0.5f*(L[i].lo[a]+L[i].hi[a]); } }
    std::vector<uint32_t> idx(n); std::iota(idx.begin(),idx.end(),0u);
    N.reserve(2*n); int32_t root=build(idx,0,(uint32_t)n);
    uint32_t limit = argc>2 ? atoi(argv[2]) : N[root].height;
    double frac = argc>3 ? atof(argv[3]) : 0.02; int passes = argc>4 ? atoi(argv[4]) : 40;
    TreeOptimiser opt(N, root, limit);
    printf("%zu leaves, height %u, cost %.4f (depth limit %u)\n", n, N[root].height, opt.cost(), limit);
    for (int p=0;p<passes;p+=5){
        auto t0=std::chrono::steady_clock::now();
        int done=opt.run(5, frac, 0.0);
        double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count();
        printf("after %d passes: cost %.4f height %u  (%.0f ms)\n", p+done, opt.cost(), N[root].height, ms);
    }
}
