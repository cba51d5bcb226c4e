// ttm_rng.h - counter-based normal deviates for the observation-noise draws of the device-resident EnTF loop
// (example_06.py:286-288 draws them with np.random: synthetic input, any generator will do - this one is stateless, so a
// draw depends only on (seed, stream, row) and the filter is reproducible whatever the launch geometry).
// Philox-4x32-10 (Salmon, Moraes, Dror, Shaw 2011) + Box-Muller on two 53-bit uniforms.
#pragma once

#include <math.h>
#include <stdint.h>

#include "ttm_vec.h"

namespace ttm {

TTM_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// standard normal deviate number `row` of stream `stream` under `seed`
TTM_HD double normal_deviate(uint64_t seed, uint32_t stream, uint64_t row) {
    uint32_t r[4];
    philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), stream, 0x5EEDu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double u1 = ((double)((((uint64_t)r[0] << 32) | r[1]) >> 11) + 0.5) * (1.0 / 9007199254740992.0);   // (0, 1)
    const double u2 = ((double)((((uint64_t)r[2] << 32) | r[3]) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586477 * u2);
}

// one RK4 step of the Lorenz-63 system (example_06.py:28-46, 49-76; sigma 10, rho 28, beta 8/3), operand order as there
TTM_HD void lorenz63_rk4_step(double& x, double& y, double& z, double dt) {
    const double beta = 8.0 / 3.0, rho = 28.0, sigma = 10.0;
    auto f = [&](double a, double b, double c, double& da, double& db, double& dc) {
        da = -sigma * a + sigma * b;
        db = -a * c + rho * a - b;
        dc = a * b - beta * c;
    };
    double k1x, k1y, k1z, k2x, k2y, k2z, k3x, k3y, k3z, k4x, k4y, k4z;
    f(x, y, z, k1x, k1y, k1z);
    f(x + dt / 2 * k1x, y + dt / 2 * k1y, z + dt / 2 * k1z, k2x, k2y, k2z);
    f(x + dt / 2 * k2x, y + dt / 2 * k2y, z + dt / 2 * k2z, k3x, k3y, k3z);
    f(x + dt * k3x, y + dt * k3y, z + dt * k3z, k4x, k4y, k4z);
    x += dt / 6 * (k1x + 2 * k2x + 2 * k3x + k4x);
    y += dt / 6 * (k1y + 2 * k2y + 2 * k3y + k4y);
    z += dt / 6 * (k1z + 2 * k2z + 2 * k3z + k4z);
}

}  // namespace ttm
