// run_planner.h — how collected path-tracing passes become pipeline runs. Host-only and HIP-free on purpose: the context
// (gpuart_hip.hip) takes every such decision here, and tests/test_run_planner.py drives the same code through
// gpuart_hip_test_planner with random sequences of resize / share / plan / mode / pass / flush on a machine without a GPU.
//
// Invariants (checked by `RunPlanner::check`, asserted again where a run is launched):
//   * every run holds 1 <= count <= max_batch passes — a lane's path buffers hold n_slots x max_batch paths, and a longer
//     run would index past them (the memory-access fault of gpurun_out/k20_plans.txt in round 2, before launch_run refused);
//   * n_slots x count <= max(batch_paths, n_slots): a run stays within the path budget unless one pass alone exceeds it;
//   * passes are launched in the order they arrived, each exactly once, and nothing stays pending after a flush.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

struct RunPlanner {
    // ---- configuration (gpuart_hip_create; GPUART_HIP_* environment) -------------------------------------------------
    uint32_t batch_limit = 64;                 ///< most passes one run may hold (MAX_BATCH, GPUART_HIP_MAX_BATCH)
    uint32_t lanes_total = 8;                  ///< pass lanes that exist (GPUART_HIP_PASSES_IN_FLIGHT)
    size_t batch_paths = (size_t)16 << 20;     ///< passes are batched while one run stays within this many paths
    size_t min_run_paths = (size_t)2 << 20;    ///< a pipeline run is not made smaller than this many paths
    size_t small_paths = (size_t)8500 << 10;   ///< mode 0: a planned sequence of at most this many paths is ONE k_run launch (0: never)
    size_t lane_budget = (size_t)32 << 30;     ///< bytes of wavefront path state over all lanes (8 lanes of 8 passes at 1080p: 17.6 GB)
    double plan_run_factor = 0;                ///< 0: as many equal runs as lanes (plan()); > 0: the rule of rounds 2-4, run length = this x sqrt(planned work) in units of 2M paths

    // ---- state ---------------------------------------------------------------------------------------------------------
    uint32_t n_slots = 0;         ///< path slots of the tile (8x8-tile padded)
    size_t tile_pixels = 0;
    uint32_t max_batch = 1;       ///< passes a lane's buffers hold
    uint32_t lanes_in_use = 1;
    uint32_t planned_passes = 0;  ///< gpuart_hip_pt_plan hint (0: unknown)
    size_t plan_done = 0;         ///< passes of the planned sequence launched so far (a sequence that completes is taken to repeat)
    int mode = 0;                 ///< gpuart_hip_set_mode
    size_t run_passes = 1;        ///< passes collected before a run starts by itself
    size_t pending = 0;           ///< passes collected, not launched yet

    static constexpr size_t PATH_BYTES = 6 * 16 + 8 + 3 * 4;  ///< path state per slot: six float4, one uint2, three queue words
    static constexpr size_t RUN_KERNEL_SLOTS = (size_t)1 << 28; ///< k_run's list entries address this many path slots of one run (kernel_run.h RUN_SLOT)

    /// Bytes of one lane's path state for runs of `batch` passes.
    size_t lane_bytes(size_t batch) const { return (size_t)n_slots * batch * PATH_BYTES + batch * tile_pixels * 16; }

    /// A new tile: the run length the budgets allow and the lanes to try first. The caller allocates `lanes_in_use` lanes of
    /// `lane_bytes(max_batch)` and calls `shrink()` while the device refuses.
    void set_tile(uint32_t slots, size_t pixels) {
        n_slots = slots; tile_pixels = pixels; pending = 0; plan_done = 0;
        if (!n_slots) { max_batch = 1; lanes_in_use = 1; run_passes = 1; return; }
        max_batch = (uint32_t)std::max<size_t>(1, std::min<size_t>({(size_t)batch_limit, batch_paths / n_slots, RUN_KERNEL_SLOTS / n_slots}));
        first_lanes();
        plan();
    }
    void first_lanes() { lanes_in_use = (uint32_t)std::min<size_t>(lanes_total, std::max<size_t>(2, lane_budget / lane_bytes(max_batch))); }
    /// The device could not give that much: fewer lanes, then shorter runs. False: nothing smaller is left.
    bool shrink() {
        if (lanes_in_use > 2) lanes_in_use = std::max<uint32_t>(2, lanes_in_use / 2);
        else if (max_batch > 1) { max_batch = (max_batch + 1) / 2; first_lanes(); }
        else if (lanes_in_use > 1) lanes_in_use = 1;
        else return false;
        plan();
        return true;
    }

    /// Whether a run of `count` passes goes through the persistent run kernel (k_run) rather than the launch pipeline.
    /// Measured on cfg3 (profiles/r02/k_run_vs_pipeline.txt): one pass alone 2.2 vs 3.4 ms, two 1.65 vs 1.98, three 1.47 vs 1.56,
    /// four 1.36 vs 1.31, 64 1.19 vs 0.93 ms per pass — k_run has no chain of dependent launches, the pipeline packs lanes better
    /// once several runs overlap. Round 6 (k_run's loop control in scalar registers: -6 %): four passes 1.10 vs 1.16, five 1.07 vs 1.01
    /// (profiles/r06/k_run_vs_pipeline_small_k.txt): the crossover moved from 3-4 to 4-5 passes of a 1080p frame, small_paths 6.4 M -> 8.5 M. Mode 0 uses k_run when the whole planned sequence is small; small_paths = 0 turns that off.
    bool uses_run_kernel(size_t count) const {
        if (mode == 1 || mode == 4 || mode == 5) return true;
        if (mode != 0 || !small_paths) return false;
        if ((size_t)n_slots * count > RUN_KERNEL_SLOTS) return false;  // (a tile beyond 2^28 pixels: one pass alone goes through the launch pipeline)
        const size_t passes = planned_passes ? planned_passes : count;
        return passes <= 1 || passes * (size_t)n_slots <= small_paths;  // one pass observed alone: k_run at every frame size
    }

    /// Passes per pipeline run of a planned sequence of K passes: AS MANY EQUAL RUNS AS THERE ARE LANES — ceil(K / lanes) passes each;
    /// a sequence longer than the lanes hold at once (lanes x max_batch) in 2, 3, ... rounds of that many runs. Measured on cfg3, K = 8 ... 48
    /// x every run length 1 ... 9 (profiles/r05/run_length_sweep.txt) and K = 64, 72, 128 (planner_lanes_rule.txt): with 8 lanes the best
    /// length is the smallest one that needs no more than 8 runs; one or two runs MORE than a multiple of the lanes is the worst a plan can
    /// do (+8 ... 15 %: the stragglers start when a lane frees up and finish with nothing to overlap with — 24 passes as 8 x 3 on the 7
    /// lanes a 16 GB budget gave: 0.95 ms per pass, on 8 lanes 0.81; 72 as 9 x 8: 0.87, as 15 x 5: 0.785), and longer runs than that rule
    /// gives pack lanes no better. Rounds 2-4 used 0.75 sqrt(work) (plan_run_factor > 0 brings it back): fitted to K = 2, 4, 8, 20, 64,
    /// right there, and in resonance with the lane count at K = 22-24, 29-32. Never below `min_run_paths` (a share's small passes),
    /// never above max_batch. Without a plan: 8M paths.
    void plan() {
        if (!n_slots) { run_passes = 1; return; }
        const double unit = (double)((size_t)2 << 20);
        const size_t min_run = std::max<size_t>(1, min_run_paths / n_slots);
        size_t want;
        if (planned_passes && (mode == 0 || mode == 5) && planned_passes <= max_batch &&
            (small_paths ? planned_passes == 1 || (size_t)planned_passes * n_slots <= small_paths : mode == 5)) {
            want = planned_passes;  // a small sequence is ONE run of the persistent kernel (uses_run_kernel)
        } else if (planned_passes && plan_run_factor > 0) {
            const double u = (double)planned_passes * n_slots / unit;
            want = (size_t)(plan_run_factor * std::sqrt(u) * unit / n_slots);
        } else if (planned_passes) {
            const size_t lanes = std::max<size_t>(1, lanes_in_use);
            const size_t rounds = (planned_passes + lanes * max_batch - 1) / (lanes * max_batch);
            want = (planned_passes + rounds * lanes - 1) / (rounds * lanes);
        } else {
            want = std::max<size_t>(1, ((size_t)8 << 20) / n_slots);
        }
        run_passes = std::min<size_t>(max_batch, std::max(min_run, want));
        // ... of EQUAL length: 20 passes in runs of 6 end in a run of 2 that has nothing left to overlap with (four runs of 5 took 5 % less
        // on a quarter-frame share, where min_run_paths sets the length)
        if (planned_passes > run_passes) {
            const size_t n_runs = (planned_passes + run_passes - 1) / run_passes;
            run_passes = std::min<size_t>(max_batch, std::max(min_run, (planned_passes + n_runs - 1) / n_runs));
        }
    }

    /// gpuart_hip_pt_plan: the caller expects `passes` passes before it observes the result (0: unknown).
    void set_plan(uint32_t passes) { planned_passes = passes; plan_done = 0; plan(); }
    void launched(size_t count) {
        plan_done += count;
        if (planned_passes && plan_done >= planned_passes) plan_done = 0;  // the sequence is complete: the next pass begins the next one
    }

    /// One more pass was collected. Returns how many pending passes start NOW as one run (0: keep collecting).
    size_t on_pass() {
        pending++;
        if (pending < run_passes && pending < max_batch) return 0;
        const size_t count = pending;
        pending = 0;
        launched(count);
        return count;
    }

    /// Everything pending must start (something observes or changes state): the run lengths, in launch order. The rest of a
    /// sequence is split over the lanes in runs of at least ~min_run_paths so that its end still overlaps; k_run fills the
    /// machine by itself and takes it as one run — but never more than a lane holds.
    std::vector<size_t> on_flush() {
        std::vector<size_t> runs;
        if (!pending) return runs;
        size_t n = uses_run_kernel(pending) ? 1 : std::max<size_t>(1, std::min<size_t>({(size_t)lanes_in_use, pending, pending * n_slots / std::max<size_t>(1, min_run_paths)}));
        // the remainder of a PLANNED sequence is the plan's last, shorter run: cut up it would take lanes the plan left alone (20 passes =
        // 6 x 3 + 2: a seventh run, not a seventh and an eighth). Only the plan's TAIL: what a read-back or a state change in the middle
        // of the sequence flushes — or passes nobody planned — is spread over the lanes as before.
        if (planned_passes && !plan_run_factor && pending <= run_passes && plan_done + pending == planned_passes) n = 1;
        n = std::max(n, (pending + max_batch - 1) / max_batch);
        for (size_t k = 0, first = 0; k < n; k++) {
            const size_t count = (pending - first) / (n - k);
            runs.push_back(count);
            first += count;
        }
        launched(pending);
        pending = 0;
        return runs;
    }

    /// nullptr if a run of `count` passes may be launched, else what is wrong with it.
    const char *check(size_t count) const {
        if (!count) return "an empty pipeline run";
        if (count > max_batch) return "a pipeline run longer than its lane's buffers";
        if ((size_t)n_slots * count > std::max<size_t>(batch_paths, n_slots)) return "a pipeline run beyond the path budget";
        return nullptr;
    }
};
