// Double integrator with friction as a TrajectoryProblem (builder-defined: the
// reference solves it by LCvx only).  Physical model from
// test/examples/double_integrator/parameters.jl:58-64: f = [x2; u - g], fixed
// duration T, so in normalised time xdot = T*[x2; u - g].  np = 0.
#pragma once
#include "model_common.hpp"

namespace scp {

struct DoubleIntegrator {
    static constexpr int id = 0;
    static constexpr int nx = 2, nu = 1, np = 0, npF = 0;
    static constexpr int npar = 2;  // [g, T]
    struct Params {
        double g, T;
    };
    static Params make_params(const double* par) { return Params{par[0], par[1]}; }
    static constexpr int Fcol(int) { return 0; }

    // f, A (col-major nx*nx), B (nx*nu), Fc (nx*npF)
    SCP_DEV static void dyn(const Params& P, double, int, const double (&x)[nx], const double (&u)[nu],
                            const double*, double (&f)[nx], double (&A)[nx * nx], double (&B)[nx * nu],
                            double (&Fc)[nx])
    {
        f[0] = P.T * x[1];
        f[1] = P.T * (u[0] - P.g);
        A[0] = 0.0; A[1] = 0.0; A[2] = P.T; A[3] = 0.0;
        B[0] = 0.0; B[1] = P.T;
        Fc[0] = 0.0; Fc[1] = 0.0;
    }
    SCP_DEV static void action(double (&)[nx]) {}
};

}  // namespace scp
