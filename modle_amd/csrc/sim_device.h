// sim_device.h -- the device code of the path: one wavefront simulates one cell, the per-epoch loop
// of Simulation::simulate_one_cell (reference: src/libmodle/cpu/simulation.cpp:896-986) written for
// a 64-lane wave.  Included after a `wave` backend (wave_hip.h on the GPU, tests/wave_emu/wave_emu.h
// on the CPU lane emulator).  The parts, in dependency order:
#pragma once
#include "sim_cell.h"              // per-cell context (Cell), phase timers, small helpers, the inverse-permutation / LDS id-filter helpers
#include "sim_pair.h"             // helper-wave mode: a second wave of the workgroup draws the moves and the barrier states of a burn-in epoch and runs the fwd instance of LEF-BAR detection (the main wave's side)
#include "sim_bind_rank.h"         // select_and_bind_lefs and rank_lefs
#include "sim_moves.h"             // generate_moves, adjust_moves_of_consecutive_extr_units, clamp_moves
#include "sim_barriers.h"          // ExtrusionBarriers::init_states / next_state and the per-epoch lists of stalling barriers
#include "sim_collisions.h"        // process_collisions: boundaries, LEF-BAR, primary and secondary LEF-LEF collisions, fix_secondary
#include "sim_release.h"           // extrude and release_lefs
#include "sim_contacts.h"          // sample_and_register_contacts
#include "sim_burnin.h"            // run_burnin: loop-size statistics and the stability test
#include "sim_helper.h"           // helper-wave mode: the helper wave's loop
#include "sim_epoch.h"             // the epoch loop of one cell (Simulation::simulate_one_cell)
#include "sim_hooks.h"             // phase-level and unit-level test entry points (the reference's Simulation::test_* hooks)
