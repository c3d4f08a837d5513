"""Small restatements used by more than one test module (test infrastructure, like oracle/)."""
import oracle as O


def minimizer_with_position(rec: bytes, m: int):
    """sequence::minimizer restated WITH the winner's window start and strand: the reference's loop order (src/sequence.rs:143-150: forward
    window i, then reverse-complement window i, i ascending, strict <) decides between equal byte strings.  Returns (bytes, start, is_rc)."""
    rcs = O.reverse_complement(rec)
    best = None
    for i in range(len(rec) - m + 1):
        for st, strand in ((0, rec), (1, rcs)):
            c = strand[i:i + m]
            if best is None or c < best[0]:
                best = (c, i, st)
    return best
