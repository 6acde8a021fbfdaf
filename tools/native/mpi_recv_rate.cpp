// Development tool: what does ONE receiving process get out of MPICH's shared-memory transport when W senders each hold a 200 MB message for it?
//   mpiexec -n 5 mpi_recv_rate        (4 senders, the last rank receives; then it sends 120 MB back to each)
// mode 0: blocking MPI_Recv one after the other (FoamYade.C:149-153's loop); 1: MPI_Irecv x W + Waitall; 2: W threads under MPI_THREAD_MULTIPLE,
// one MPI_Recv each; 3: 16 chunks per sender.  MPICH 3.3.2 ch3:nemesis, this container (8 vCPU): 5.3 - 5.5 GB/s in EVERY mode -- the receiving
// process's one core copies every byte out of the shared-memory cells and the progress engine runs under one lock, so neither non-blocking
// receives nor threads add anything (the GPU boxes' EPYC: 8.9 GB/s).  Hence the wire helpers of include/foamyade_mpi.h: several receiving PROCESSES.
#include <mpi.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <cstring>
#include <thread>
static double now(){return std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();}
int main(int argc,char**argv){
  int prov; MPI_Init_thread(&argc,&argv,MPI_THREAD_MULTIPLE,&prov);
  int r,w; MPI_Comm_rank(MPI_COMM_WORLD,&r); MPI_Comm_size(MPI_COMM_WORLD,&w);
  const int W=w-1; const size_t n=25000000; // 200 MB doubles
  const int root=w-1;
  if(r!=root){ std::vector<double> b(n,1.0);
    for(int mode=0;mode<4;++mode) for(int rep=0;rep<3;++rep){ MPI_Barrier(MPI_COMM_WORLD);
      if(mode<3) MPI_Send(b.data(),(int)n,MPI_DOUBLE,root,7,MPI_COMM_WORLD);
      else { // chunked 
        const int C=16; for(int c=0;c<C;++c) MPI_Send(b.data()+c*(n/C),(int)(n/C),MPI_DOUBLE,root,7,MPI_COMM_WORLD);} 
      if (mode<3) MPI_Recv(b.data(),(int)(n*6/10),MPI_DOUBLE,root,8,MPI_COMM_WORLD,MPI_STATUS_IGNORE);
    }
  } else { if(r==root) printf("thread level provided %d (MULTIPLE=%d)\n",prov,MPI_THREAD_MULTIPLE);
    std::vector<std::vector<double>> b(W,std::vector<double>(n)); 
    for(int mode=0;mode<4;++mode) for(int rep=0;rep<3;++rep){ MPI_Barrier(MPI_COMM_WORLD); double t0=now();
      if(mode==0){ for(int s=0;s<W;++s) MPI_Recv(b[s].data(),(int)n,MPI_DOUBLE,s,7,MPI_COMM_WORLD,MPI_STATUS_IGNORE);}
      else if(mode==1){ std::vector<MPI_Request> rq(W); for(int s=0;s<W;++s) MPI_Irecv(b[s].data(),(int)n,MPI_DOUBLE,s,7,MPI_COMM_WORLD,&rq[s]); MPI_Waitall(W,rq.data(),MPI_STATUSES_IGNORE);}
      else if(mode==2){ std::vector<std::thread> th; for(int s=0;s<W;++s) th.emplace_back([&,s]{MPI_Recv(b[s].data(),(int)n,MPI_DOUBLE,s,7,MPI_COMM_WORLD,MPI_STATUS_IGNORE);}); for(auto&t:th)t.join(); }
      else { const int C=16; for(int s=0;s<W;++s) for(int c=0;c<C;++c) MPI_Recv(b[s].data()+c*(n/C),(int)(n/C),MPI_DOUBLE,s,7,MPI_COMM_WORLD,MPI_STATUS_IGNORE);} 
      double t1=now();
      double ts=0;
      if(mode<3){ double t2=now();
        if(mode==0) for(int s=0;s<W;++s) MPI_Send(b[s].data(),(int)(n*6/10),MPI_DOUBLE,s,8,MPI_COMM_WORLD);
        else { std::vector<MPI_Request> rq(W); for(int s=0;s<W;++s) MPI_Isend(b[s].data(),(int)(n*6/10),MPI_DOUBLE,s,8,MPI_COMM_WORLD,&rq[s]); MPI_Waitall(W,rq.data(),MPI_STATUSES_IGNORE);} 
        ts=now()-t2; }
      printf("mode %d rep %d: recv %d x 200 MB in %.1f ms = %.1f GB/s ; send %d x 120 MB in %.1f ms = %.1f GB/s\n",mode,rep,W,t1-t0,W*n*8/(t1-t0)*1e-6,W,ts,ts>0?W*n*4.8/ts*1e-6:0.0);
    }
  }
  MPI_Finalize(); }
