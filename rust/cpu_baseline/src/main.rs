//! needletail's own per-record chain on bench.py's reads: normalize -> reverse_complement -> canonical_kmers
//! (reference src/sequence.rs:226-239, src/kmer.rs:48-130), reduced to the scalars bench.py compares with the GPU result.
//!     ntk_cpu_baseline <reads file> <n_reads> <read_len> <k> <threads>
//! The file holds n_reads records of read_len bytes, each followed by one separator byte.  Prints one JSON line.
use needletail::Sequence;
use std::time::Instant;

#[derive(Default, Clone, Copy)]
struct Stats { n_total: u64, n_fwd: u64, sum: u64, xor: u64 }

fn value_of(kmer: &[u8]) -> u64 {
    // the 2-bit value of a yielded slice: A0 C1 G2 T3, first base most significant (reference src/bitkmer.rs:5-36)
    kmer.iter().fold(0u64, |v, &b| (v << 2) | match b { b'A' | b'a' => 0, b'C' | b'c' => 1, b'G' | b'g' => 2, _ => 3 })
}

fn run(records: &[u8], stride: usize, read_len: usize, k: u8) -> Stats {
    let mut s = Stats::default();
    for rec in records.chunks_exact(stride) {
        let seq = &rec[..read_len];
        let norm = seq.normalize(false);
        let rc = norm.reverse_complement();
        for (_pos, kmer, is_rc) in norm.canonical_kmers(k, &rc) {
            let v = value_of(kmer);
            s.n_total += 1;
            s.n_fwd += (!is_rc) as u64;
            s.sum = s.sum.wrapping_add(v);
            s.xor ^= v;
        }
    }
    s
}

fn main() {
    let a: Vec<String> = std::env::args().collect();
    let data = std::fs::read(&a[1]).expect("reads file");
    let (n_reads, read_len, k, threads): (usize, usize, u8, usize) =
        (a[2].parse().unwrap(), a[3].parse().unwrap(), a[4].parse().unwrap(), a[5].parse().unwrap());
    let stride = read_len + 1;
    let data = &data[..n_reads * stride];
    let t0 = Instant::now();
    let parts: Vec<Stats> = std::thread::scope(|sc| {
        let hs: Vec<_> = (0..threads).map(|t| {
            let (r0, r1) = (n_reads * t / threads, n_reads * (t + 1) / threads);
            let d = &data[r0 * stride..r1 * stride];
            sc.spawn(move || run(d, stride, read_len, k))
        }).collect();
        hs.into_iter().map(|h| h.join().unwrap()).collect()
    });
    let secs = t0.elapsed().as_secs_f64();
    let mut s = Stats::default();
    for p in parts { s.n_total += p.n_total; s.n_fwd += p.n_fwd; s.sum = s.sum.wrapping_add(p.sum); s.xor ^= p.xor; }
    println!("{{\"n_total\": {}, \"n_fwd\": {}, \"sum\": {}, \"xor\": {}, \"seconds\": {}}}", s.n_total, s.n_fwd, s.sum, s.xor, secs);
}
