// Compile-and-link check of include/hot_adapter.hpp against libhotmi355x.so (run by tests/test_abi_load.py).
// Without a GPU the constructor must throw (no CPU fallback); with one it advances a tiny cloud.
#include "hot_adapter.hpp"
#include <cstdio>
#include <vector>
int main()
{
    hot_config cfg = hotmi::Simulation<double>::defaults();
    cfg.levelCnt = 2;
    try {
        hotmi::Simulation<double> sim(cfg);
        hotmi::Objective<double> obj(sim);
        const int n = 4, ppc = 8;
        std::vector<double> X, V, m, vol, mu, la;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                for (int k = 0; k < n; ++k)
                    for (int q = 0; q < ppc; ++q) {
                        double o[3] = { 0.25 + 0.5 * (q & 1), 0.25 + 0.5 * ((q >> 1) & 1), 0.25 + 0.5 * (q >> 2) };
                        X.insert(X.end(), { 5 + (i + o[0]) * cfg.dx, 5 + (j + o[1]) * cfg.dx, 5 + (k + o[2]) * cfg.dx });
                        V.insert(V.end(), { 0.0, -0.1, 0.0 });
                        m.push_back(2000 * cfg.dx * cfg.dx * cfg.dx / ppc), vol.push_back(cfg.dx * cfg.dx * cfg.dx / ppc), mu.push_back(19230.77), la.push_back(28846.15);
                    }
        sim.setParticles((int64_t)m.size(), X.data(), V.data(), m.data(), nullptr, nullptr, vol.data(), mu.data(), la.data());
        sim.advanceOneTimeStep(1.0 / 24);
        std::printf("adapter ok: %d nodes, %d iterations, converged %d\n", sim.stats.num_nodes, sim.stats.iterations, sim.stats.converged);
        return sim.stats.converged ? 0 : 1;
    }
    catch (const std::exception& e) {
        std::printf("adapter threw: %s\n", e.what());
        return 42;
    }
}
