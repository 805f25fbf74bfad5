// Uses include/openpbso_amd_facade.h the way tools/real_time_modal_sound.cpp
// uses modal_solver.h: BuildSolver (:309-345), a Shift+click hit
// (GetModalForceVertex :268-295 + enqueueForceMessage :610), the simulation
// thread's step() loop (:527-536) and PaModalCallback (:192-212).  Writes the
// mono float32 stream the callback would play to argv[1].
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "openpbso_amd_facade.h"

template <typename T>
struct ModeDataLite {                       // ModeData<T> accessors used by the tool
    std::vector<T> _omegaSquared;
    std::vector<std::vector<T>> _modes;
    const std::vector<T> &mode(int i) const { return _modes.at(i); }
    int numModes() const { return (int)_omegaSquared.size(); }
};

// tools/real_time_modal_sound.cpp:268-295, verbatim control flow with the facade types
template <typename T>
void GetModalForceVertex(const int forceDim, const ModeDataLite<T> &modes, const int vid, const double vn[3],
                         ForceMessage<double> &force, ForceType type, float gaussWidth) {
    force.data.setZero(forceDim);
    for (int mm = 0; mm < forceDim; ++mm)
        force.data(mm) = vn[0] * modes.mode(mm).at(vid * 3 + 0) + vn[1] * modes.mode(mm).at(vid * 3 + 1) +
                         vn[2] * modes.mode(mm).at(vid * 3 + 2);
    force.forceType = type;
    if (type == ForceType::PointForce) force.force.reset(new PointForce<T>());
    else if (type == ForceType::GaussianForce) force.force.reset(new GaussianForce<T>(gaussWidth));
    else force.force.reset(new AutoregressiveForce<T>());
}

struct PaModalData {
    std::unique_ptr<ModalSolver<double>> *solver;
    SoundMessage<double> soundMessage;
};
// tools/real_time_modal_sound.cpp:192-212
static int PaModalCallback(void *outputBuffer, unsigned long framesPerBuffer, void *userData) {
    PaModalData *data = (PaModalData *)userData;
    float *out = (float *)outputBuffer;
    (*(data->solver))->dequeueSoundMessage(data->soundMessage);
    for (unsigned i = 0; i < framesPerBuffer; i++) {
        *out++ = (float)(data->soundMessage.data(i) / 1E10);
        *out++ = (float)(data->soundMessage.data(i) / 1E10);
    }
    return 0;
}

// the lock-free sound / qnorm queue of the facade: capacity and FIFO order under a real producer / consumer pair
static int ring_selftest() {
    pbso_facade::SpscRing<int> r;
    int x = -1;
    if (r.try_dequeue(x)) return 10;
    for (int i = 0; i < 3; ++i) if (!r.try_enqueue(i)) return 11;      // ReaderWriterQueue(2): 3 usable slots
    if (r.try_enqueue(3)) return 12;                                   // the fourth try_enqueue fails (queue full)
    if (r.size_approx() != 3) return 13;
    for (int i = 0; i < 3; ++i) if (!r.try_dequeue(x) || x != i) return 14;
    if (r.try_dequeue(x)) return 15;
    const int n = 200000;
    long long sum = 0;
    bool ordered = true;
    std::thread consumer([&]() {
        int want = 0, v = 0;
        while (want < n) {
            if (r.try_dequeue(v)) { ordered = ordered && v == want; sum += v; ++want; }
        }
    });
    for (int i = 0; i < n; ++i) while (!r.try_enqueue(i)) {}
    consumer.join();
    if (!ordered || sum != (long long)n * (n - 1) / 2) return 16;
    std::printf("ring selftest ok\n");
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "--ring-selftest") return ring_selftest();
    const char *out_path = argc > 1 ? argv[1] : "facade_out.f32";
    const int n_modes = 96, n_verts = 8, n_buffers = 6;
    // deterministic "model": eigenvalues 200 Hz .. 9 kHz, mode shapes from a LCG
    ModeDataLite<double> modes;
    unsigned lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return ((lcg >> 8) & 0xFFFF) / 65536.0 - 0.5; };
    for (int m = 0; m < n_modes; ++m) {
        const double f = 200.0 * std::pow(45.0, (double)m / (n_modes - 1));
        modes._omegaSquared.push_back(2500.0 * std::pow(2 * M_PI * f, 2));
        std::vector<double> u(3 * n_verts);
        for (auto &x : u) x = rnd() * 2e-3;
        modes._modes.push_back(u);
    }
    // BuildSolver
    std::unique_ptr<ModalSolver<double>> solver(new ModalSolver<double>(n_modes));
    std::shared_ptr<ModalIntegrator<double>> integrator(ModalIntegrator<double>::Build(
        2500.0, modes._omegaSquared, 6.0, 1e-7, 1. / (double)SAMPLE_RATE, n_modes));
    solver->setIntegrator(integrator);
    solver->setUseTransfer(false);                   // no FFAT maps in this smoke model
    // a Shift+click on vertex 3, a Gaussian-force hit two buffers later
    const double vn[3] = {0.6, 0.0, 0.8};
    ForceMessage<double> force;
    GetModalForceVertex(n_modes, modes, 3, vn, force, ForceType::PointForce, 0.f);
    if (!solver->enqueueForceMessage(force)) return 2;
    PaModalData pa;
    pa.solver = &solver;
    std::vector<float> stereo(2 * FRAMES_PER_BUFFER), mono;
    for (int b = 0; b < n_buffers; ++b) {
        if (b == 2) {
            GetModalForceVertex(n_modes, modes, 5, vn, force, ForceType::GaussianForce, 300.f);
            solver->enqueueForceMessageNoFail(force);
        }
        solver->step();                              // simulation thread
        PaModalCallback(stereo.data(), FRAMES_PER_BUFFER, &pa);   // audio thread
        for (int i = 0; i < FRAMES_PER_BUFFER; ++i) mono.push_back(stereo[2 * i]);
    }
    // the GUI's "Clear force" button (tools/real_time_modal_sound.cpp:745-747): a default-constructed message,
    // clearAllForces = true, NO data; that step() emits no buffer (modal_solver.h:186-189), the next one does
    {
        ForceMessage<double> clear;
        clear.clearAllForces = true;
        const bool ok = solver->enqueueForceMessageNoFail(clear, 4);
        solver->step();
        SoundMessage<double> none;
        const bool sound = solver->dequeueSoundMessage(none);
        std::printf("clear: enqueued=%d sound_after_clear=%d\n", (int)ok, (int)sound);
        solver->step();
        PaModalCallback(stereo.data(), FRAMES_PER_BUFFER, &pa);
        for (int i = 0; i < FRAMES_PER_BUFFER; ++i) mono.push_back(stereo[2 * i]);
    }
    // a missing FFAT directory is an EMPTY map (io.cpp:31-34 + LoadAll): nothing happens until computeTransfer
    // asks for _ffat_maps->at(0), which throws std::out_of_range (modal_solver.h:294)
    {
        std::unique_ptr<ModalSolver<double>> s2(new ModalSolver<double>(4));
        std::vector<double> om(modes._omegaSquared.begin(), modes._omegaSquared.begin() + 4);
        s2->setIntegrator(std::shared_ptr<ModalIntegrator<double>>(ModalIntegrator<double>::Build(2500.0, om, 6.0, 1e-7, 1. / SAMPLE_RATE, 4)));
        s2->readFFATMaps("/nonexistent_ffat_dir_of_the_facade_test");
        s2->step();                                   // fine: the unit transfer is in effect
        bool threw = false;
        try {
            pbso_facade::VecN<double, 3> pos;
            pos(0) = 1; pos(1) = 2; pos(2) = 3;
            s2->computeTransfer(pos);
        } catch (const std::out_of_range &) {
            threw = true;
        }
        std::printf("missing_ffat_dir: out_of_range=%d\n", (int)threw);
    }
    // two AR-parameter updates back to back from a second thread while the simulation thread steps (modal_solver.h:382-393): the
    // second finds the 1-slot queue full and spins -- without holding the engine's lock between attempts -- until a step() of a
    // sustained AutoregressiveForce contact (:226-236) has taken the first; with a bounded maxIte and nobody stepping it gives up
    {
        std::unique_ptr<ModalSolver<double>> s3(new ModalSolver<double>(n_modes));
        s3->setIntegrator(integrator);
        s3->setUseTransfer(false);
        ForceMessage<double> f;
        GetModalForceVertex(n_modes, modes, 2, vn, f, ForceType::AutoregressiveForce, 0.f);
        f.sustainedForceStart = true;
        const bool started = s3->enqueueForceMessage(f);
        SoundMessage<double> sm;
        s3->step();                                   // the contact is on
        s3->dequeueSoundMessage(sm);
        AutoregressiveForceParam<double> p1, p2;
        p1.sigma = 0.002;
        p2.sigma = 0.003;
        const bool first = s3->enqueueArprmMessageNoFail(p1, 5);      // empty slot: accepted at once
        const bool bounded = s3->enqueueArprmMessageNoFail(p2, 5);    // full slot, nobody steps: false after five tries
        std::atomic<int> steps_started{0};
        std::atomic<bool> done{false};
        bool second = false;
        int steps_at_accept = -1;
        std::thread gui([&]() {
            second = s3->enqueueArprmMessageNoFail(p2);               // maxIte = -1: spins until accepted
            steps_at_accept = steps_started.load();
            done = true;
        });
        int guard = 0;
        while (!done && guard++ < 2000) {
            ++steps_started;
            s3->step();
            s3->dequeueSoundMessage(sm);
        }
        gui.join();
        std::printf("arprm: started=%d first=%d bounded=%d second=%d after_a_step=%d\n", (int)started, (int)first, (int)bounded, (int)second,
                    (int)(steps_at_accept >= 1));
    }
    pbso_facade::VecX<double> qn = solver->getQBufferNorm();
    const TransMessage<double> &tr = solver->getLatestTransfer();
    std::printf("qnorm[0]=%g transfer[0]=%g n=%d\n", qn(0), tr.data(0), tr.data.size());
    FILE *f = std::fopen(out_path, "wb");
    if (!f) return 3;
    std::fwrite(mono.data(), sizeof(float), mono.size(), f);
    std::fclose(f);
    // also dump the model so the test can rebuild it for the oracle
    std::string mp = std::string(out_path) + ".model";
    f = std::fopen(mp.c_str(), "wb");
    std::fwrite(modes._omegaSquared.data(), sizeof(double), n_modes, f);
    for (int m = 0; m < n_modes; ++m) std::fwrite(modes._modes[m].data(), sizeof(double), 3 * n_verts, f);
    std::fclose(f);
    return 0;
}
