// rtcuda/profiler.hpp -- stage timing lines on stdout, in the format of the reference's driver (profiler.hpp:14-28):
//     "<name>... done (<milliseconds>ms)"
// so the log of a run lines up with the reference's, stage by stage.  Same interface --
// `profiler.start(name)` / `profiler.stop()` on one global object -- plus a scope guard; `profiler.enabled = false`
// silences it (library code only prints through it, and only when asked to).
#ifndef RTCUDA_HOST_PROFILER_HPP
#define RTCUDA_HOST_PROFILER_HPP

#include <cassert>
#include <chrono>
#include <iostream>
#include <string>

struct Profiler {
    using clock = std::chrono::steady_clock;
    bool enabled = true;
    bool running = false;
    clock::time_point start_time;
    float last_ms = 0.f;  // duration of the stage that ended last

    void start(const std::string &name) {
        assert(!running);
        running = true;
        if (enabled) std::cout << name << "... " << std::flush;
        start_time = clock::now();
    }
    void stop() {
        const clock::time_point end_time = clock::now();
        assert(running);
        running = false;
        last_ms = std::chrono::duration<float, std::milli>(end_time - start_time).count();
        if (enabled) std::cout << "done (" << last_ms << "ms)" << std::endl;
    }
    // `Profiler::Stage s(profiler, "Rendering");` = start now, stop at the end of the scope (also when it is left by an exception)
    struct Stage {
        Profiler &p;
        Stage(Profiler &profiler_, const std::string &name) : p(profiler_) { p.start(name); }
        ~Stage() { if (p.running) p.stop(); }
        Stage(const Stage &) = delete;
        Stage &operator=(const Stage &) = delete;
    };
};

inline Profiler profiler;  // (C++17 inline variable: one object however many translation units include this)

#endif  // RTCUDA_HOST_PROFILER_HPP
