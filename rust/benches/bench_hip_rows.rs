// Rows to add to the reference's criterion harness (benches/bench_n_to_bits.rs) so the CPU and
// GPU paths are driven side by side.  NOT COMPILED here (no Rust toolchain, criterion not
// vendored); bench/bench_n_to_bits.cpp is the executable C++ twin of exactly these rows.
//
// in bench_n_to_bits():    group.bench_function("n_to_bits_hip",  |b| b.iter(|| n_to_bits_hip(&n)));
// in bench_bits_to_n():    group.bench_function("bits_to_n_hip",  |b| b.iter(|| bits_to_n_hip(&bits, len)));
// in bench_n_to_bits2():   group.bench_function("n_to_bits2_hip", |b| b.iter(|| n_to_bits2_hip(&n)));
// in bench_bits_to_n2():   group.bench_function("bits_to_n2_hip", |b| b.iter(|| bits_to_n2_hip(&bits, len)));
//
// with `use cute_nucleotides::hip::*;` next to the existing `use` lines (:3-4).
