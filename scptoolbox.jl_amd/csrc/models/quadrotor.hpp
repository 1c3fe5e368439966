// Quadrotor obstacle avoidance dynamics, test/examples/quadrotor/definition.jl:140-186:
// x=[r;v], u=[a;sigma], p=[tdil]; f=[v; a+g]*tdil, A[r,v]=I*tdil, B[v,a]=I*tdil,
// F[:,tdil] = f/tdil.  Gravity g=(0,0,-gnrm), parameters.jl:58-60,109.
#pragma once
#include "model_common.hpp"

namespace scp {

struct Quadrotor {
    static constexpr int id = 1;
    static constexpr int nx = 6, nu = 4, np = 1, npF = 1;
    static constexpr int npar = 1;  // [gnrm]
    struct Params {
        double gnrm;
    };
    static Params make_params(const double* par) { return Params{par[0]}; }
    static constexpr int Fcol(int) { return 0; }

    SCP_DEV static void dyn(const Params& P, double, int, const double (&x)[nx], const double (&u)[nu],
                            const double* p, double (&f)[nx], double (&A)[nx * nx], double (&B)[nx * nu],
                            double (&Fc)[nx])
    {
        const double tdil = p[0];
        double f0[nx];
        f0[0] = x[3]; f0[1] = x[4]; f0[2] = x[5];
        f0[3] = u[0]; f0[4] = u[1]; f0[5] = u[2] - P.gnrm;
#pragma unroll
        for (int i = 0; i < nx; i++) f[i] = f0[i] * tdil;
        zero(A);
        A[0 + nx * 3] = tdil; A[1 + nx * 4] = tdil; A[2 + nx * 5] = tdil;
        zero(B);
        B[3 + nx * 0] = tdil; B[4 + nx * 1] = tdil; B[5 + nx * 2] = tdil;
#pragma unroll
        for (int i = 0; i < nx; i++) Fc[i] = f[i] / tdil;  // definition.jl:180
    }
    SCP_DEV static void action(double (&)[nx]) {}
};

}  // namespace scp
