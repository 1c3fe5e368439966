// Mars rocket landing as a free-final-time TrajectoryProblem (builder-defined,
// SURVEY.md F6) over the physical model of
// test/examples/rocket_landing/parameters.jl:77-146:
//   x=[r;v;z=ln m], u=[a;xi], p=[tf];  xdot = tf*(A_c x + B_c u + p_c)
//   A_c = [0 I 0; -(w^x)^2 -2 w^x 0; 0 0 0], B_c = [0; I 0; 0 -alpha], p_c=[0;g;0]
#pragma once
#include "model_common.hpp"

namespace scp {

struct RocketLanding {
    static constexpr int id = 2;
    static constexpr int nx = 7, nu = 4, np = 1, npF = 1;
    static constexpr int npar = 7;  // [g(3), omega(3), alpha]
    struct Params {
        double g[3];
        double S2[9];  // -(w^x)^2, col-major
        double S[9];   // -2 w^x,   col-major
        double alpha;
    };
    static Params make_params(const double* par)
    {
        Params P;
        const double* w = par + 3;
        const double K[9] = {0, w[2], -w[1], -w[2], 0, w[0], w[1], -w[0], 0};  // skew(w) col-major
        for (int i = 0; i < 3; i++) P.g[i] = par[i];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double a = 0;
                for (int l = 0; l < 3; l++) a += K[i + 3 * l] * K[l + 3 * j];
                P.S2[i + 3 * j] = -a;
                P.S[i + 3 * j] = -2.0 * K[i + 3 * j];
            }
        P.alpha = par[6];
        return P;
    }
    static constexpr int Fcol(int) { return 0; }

    SCP_DEV static void dyn(const Params& P, double, int, const double (&x)[nx], const double (&u)[nu],
                            const double* p, double (&f)[nx], double (&A)[nx * nx], double (&B)[nx * nu],
                            double (&Fc)[nx])
    {
        const double tf = p[0];
        zero(A);
#pragma unroll
        for (int i = 0; i < 3; i++) A[i + nx * (3 + i)] = 1.0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                A[(3 + i) + nx * j] = P.S2[i + 3 * j];
                A[(3 + i) + nx * (3 + j)] = P.S[i + 3 * j];
            }
        double f0[nx];
#pragma unroll
        for (int i = 0; i < nx; i++) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < nx; j++) a += A[i + nx * j] * x[j];
            f0[i] = a;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) f0[3 + i] += u[i] + P.g[i];
        f0[6] += -P.alpha * u[3];
#pragma unroll
        for (int i = 0; i < nx; i++) f[i] = f0[i] * tf;
#pragma unroll
        for (int i = 0; i < nx * nx; i++) A[i] *= tf;
        zero(B);
        B[3 + nx * 0] = tf; B[4 + nx * 1] = tf; B[5 + nx * 2] = tf;
        B[6 + nx * 3] = -P.alpha * tf;
#pragma unroll
        for (int i = 0; i < nx; i++) Fc[i] = f[i] / tf;
    }
    SCP_DEV static void action(double (&)[nx]) {}
};

}  // namespace scp
