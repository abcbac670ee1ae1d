// How many errbd / truncation evaluations, cutoff-search doublings and bisections one Davies evaluation spends before its main
// integration (host build of rvt_davies.h with the RVT_DV_PROFILE counters), over synthetic coefficient sets and a sweep of points.
// g++ -O2 -o /tmp/davies_calls tools/davies_calls.cpp && /tmp/davies_calls
#define RVT_DV_PROFILE
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
#include <cmath>
namespace rvt { long long rvt_dv_profile[8]; }
#include "../rvtests_amd/csrc/rvt_davies.h"
using namespace rvt;
int main(){
  std::mt19937_64 rng(5);
  std::gamma_distribution<double> gam(1.0, 1.0);
  const int M=50, NS=20, NC=400;
  long long tot[8]={0}; long long evals=0, direct=0, nmain=0; double terms=0;
  for(int s=0;s<NS;++s){
    std::vector<double> lam(M); for(auto& l:lam) l=gam(rng)*std::exp(3.0*((double)rng()/1.8e19-0.5));
    std::sort(lam.begin(),lam.end(),std::greater<double>());
    std::vector<int> th(M); davies_order(lam.data(),M,th.data());
    DaviesPrelude pre; davies_prelude(lam.data(),th.data(),M,10000,1e-6,&pre,true);
    double mu=0,var=0; for(double l:lam){mu+=l;var+=2*l*l;}
    for(auto&x:rvt_dv_profile)x=0;
    for(int i=0;i<NC;++i){
      double c = (mu+4*std::sqrt(var))*(i+0.5)/NC;
      DaviesTask task; davies_qf_front(lam.data(),th.data(),M,c,10000,1e-6,&pre,&task,true);
      ++evals; if(task.need_main){++nmain; terms+=task.nt+1;} terms+=task.nterms;
    }
    for(int k=0;k<8;++k) tot[k]+=rvt_dv_profile[k];
  }
  printf("evals %lld need_main %lld  per eval: errbd %.2f trunc %.2f doublings %.2f bisections %.2f auxint %.3f  terms/eval %.1f\n",evals,nmain,
    (double)tot[0]/evals,(double)tot[1]/evals,(double)tot[2]/evals,(double)tot[3]/evals,(double)tot[4]/evals, terms/evals);
}
