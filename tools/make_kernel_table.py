#!/usr/bin/env python3
"""DESIGN.md section 4's kernel table from two rocprofv3 --kernel-trace --stats summaries of bench.py runs (config 3 on one GPU, a C3 shard):
one row per kernel of config 3's top N by time (and of the shard's top 12), ms per step at both sizes, the family, what it does and the resource that bounds it.
usage: make_kernel_table.py <c3full kernel_stats.csv> <steps> <c3shard kernel_stats.csv> <steps> [N=40]
(steps = timed + warm-up + the one untimed pass with events: every kernel of a step runs that many times in the profile)"""
import csv
import sys

INFO = {  # kernel (prefix) -> (family, what, bound)
    "gl_apply_k<3, true": ("gram / project / kmpp", "LDS-banded pass, 10-column panel (both Gram passes, the wide / thin products)", "LDS issue + HBM ids"),
    "gl_apply_k<2": ("project / kmpp / sparse", "the same pass on 6- / 8-column panels", "LDS issue + HBM ids"),
    "gl_apply_k<1": ("kmpp / movers", "the same pass on 2- / 4-column panels", "LDS issue + HBM ids"),
    "gl_apply_k<3, false": ("project", "12-column panel", "LDS issue + HBM ids"),
    "gl_reduce_cm_k": ("gram", "pass 2's slabs summed in fixed order, scaled by s_w, written column-major", "HBM"),
    "gl_pack_scale_k": ("gram", "diag(s) X packed as the planar band image", "launch"),
    "gl_pack_panel_k": ("project / kmpp", "a panel of a wide operand packed as the band image", "launch"),
    "gl_wide_assemble_k": ("project", "16 panels' position-ordered rows -> document-major rows of P, norms, the bf16 split copy", "HBM"),
    "vtf_mfma_k": ("ortho", "H = V^T F, v_mfma_f32_16x16x4_f32, panel chunk staged in LDS per 8-wave workgroup, fp64 across row chunks", "HBM (reads the basis at 5.4 TB/s; before round 6: the vector L1 tag path)"),
    "vtf_reduce_k": ("ortho", "row-chunk partials of H summed in fixed order", "launch"),
    "vtf_partial_k": ("ortho", "H = V^T F for narrow bases (FMA form)", "HBM"),
    "update_mfma_k": ("ortho", "F -= V H on the matrix cores, a 16-byte coefficient load per 16 MFMAs", "HBM (reads the basis)"),
    "pqr_gram_k": ("qr", "slab Gram matrices of the panel (fp64)", "latency"),
    "pqr_factor_k": ("qr", "sum of the slab Grams, Cholesky with column dropping, triangular inverse (one workgroup; wave 0)", "latency"),
    "pqr_apply_gram_k": ("qr", "Q1 = F T and the slab Grams of Q1 in one sweep", "latency"),
    "pqr_apply_k": ("qr", "Q = Q1 T2", "latency"),
    "td_persist_k": ("evd", "Householder tridiagonalisation, matrix resident in LDS, one grid barrier per column", "latency chain: n x ~10 us"),
    "td_back_k": ("evd", "Z = Q Z_T, reflectors applied one by one (ISLE_TD_BACK=seq)", "latency: 2 barriers per reflector"),
    "td_back_wy_k": ("evd", "Z = Q Z_T by blocks of four reflectors in compact WY form, the block's columns staged by LDS-DMA one block ahead", "latency: 2 barriers per four reflectors"),
    "td_wy_T_k": ("evd", "the 4 x 4 T factors of the reflector blocks", "launch"),
    "td_bisect_k": ("evd", "eigenvalues by 64-way multisection on the Sturm count", "latency"),
    "td_vectors_k": ("evd", "eigenvectors of T by twisted factorisation, a thread per vector", "latency"),
    "td_check_k": ("evd", "orthogonality of neighbouring vectors", "launch"),
    "td_sym_k": ("evd", "fp32 input -> symmetric fp64 work matrix", "launch"),
    "isle_gemm::gemm_f32_k<isle_gemm::Cfg<2, 2, 4, 4, 16, 4>": ("rotate / lift", "plain f32 GEMM 256x256 tiles, v_mfma_f32_32x32x2_f32 (Ritz rotation, lift)", "MFMA f32"),
    "isle_gemm::gemm_f32_k<isle_gemm::Cfg<2, 1": ("kmpp / movers", "thin f32 GEMM (W = U C_new^T)", "MFMA f32 / HBM"),
    "isle_gemm::gemm_f32_k": ("dense", "plain f32 GEMM, other tile shapes", "MFMA f32"),
    "isle_gemm3::gemm_bf16x2_dma_k<isle_gemm3::CfgDma<2, 16>, YyGroupEpi": ("sparse", "first assignment of Lloyd on B: D x k x k on two bf16 terms by LDS-DMA, Yinyang group epilogue", "MFMA bf16"),
    "isle_gemm3::gemm_bf16x2_dma_k<isle_gemm3::CfgDma<2, 16>, TileEpi": ("lloyd_proj", "full assignment pass of Lloyd in span(U), same product, tile-bound epilogue", "MFMA bf16"),
    "isle_gemm3::gemm_bf16x3_k<isle_gemm3::Cfg<2, 2, 4, 4, 4, 16, 2>": ("lloyd_proj", "two-term product on the gathered rows of the active documents", "MFMA bf16"),
    "isle_gemm3::gemm_bf16x3_k<isle_gemm3::Cfg<2, 2, 4, 4, 4, 16, 3>": ("lloyd_proj / sparse", "three-term product on the rows the two-term pass left open", "MFMA bf16"),
    "gemm2_split_a_k": ("project", "P split into two bf16 terms (LDS image per row block and slab)", "HBM"),
    "yy2_filter_tighten_k": ("sparse", "Yinyang by group: bounds lowered by the group movements, active documents tightened", "HBM (group bounds)"),
    "yy2_scan_k": ("sparse", "(group, document) pairs scanned in group order from a 3.2 MB table in L2", "L2 round trips"),
    "yy2_pack_k": ("sparse", "centres copied group-major", "HBM"),
    "yy2_emit_k": ("sparse", "pair lists", "HBM"),
    "yy2_combine_k": ("sparse", "a document's scans folded in group order", "HBM"),
    "yy_first_combine_k": ("sparse", "winner per row from the product's candidate records", "HBM"),
    "tiles_combine_k": ("lloyd_proj", "winner per row from the product's candidate records", "HBM"),
    "pt_filter_k": ("lloyd_proj", "tile bounds lowered, candidates found", "HBM"),
    "pt_tighten_ahead_k": ("lloyd_proj", "exact distance to the own centre for the candidates", "HBM"),
    "compact_rows64_k": ("lloyd_proj", "active documents' rows gathered coordinate-major", "HBM"),
    "compact_rows_k": ("lloyd_proj / sparse", "row compaction", "HBM"),
    "proj_segsum_k": ("lloyd_proj", "centroid sums over member lists in fixed order", "HBM"),
    "proj_delta_sum_k": ("lloyd_proj", "centroid sums kept up to date by the documents that moved", "HBM"),
    "proj_changed_k": ("lloyd_proj", "documents that changed centre", "HBM"),
    "kmpp_min_dots_track_k": ("kmpp", "running minimum of the distances to the new seeds, nearest seed / tile minima kept", "HBM"),
    "kmpp_to_tiles_k": ("kmpp", "hand-over of the tracked state to Lloyd's tile bounds", "HBM"),
    "isle_scan::scan": ("kmpp", "fp64 prefix sums of D^2 (reduce / final)", "HBM"),
    "colnorm_partial_k": ("ortho / qr", "column norms (residual rule, rank repair)", "HBM"),
    "rownorms_k": ("kmeans", "squared norms of centre rows", "launch"),
    "scale_centers_k": ("lloyd_proj / sparse", "sums -> means", "launch"),
    "cc_hist_k": ("sparse_update", "word histogram of 2048 member entries in LDS, one integer atomic per word present", "LDS integer atomics"),
    "cc_centers_k": ("sparse_update", "count table -> row-major centres", "HBM"),
    "cc_moved_k": ("sparse_update", "counts updated by the documents that moved", "HBM"),
    "doc_norms_k": ("sparse", "squared norms of the documents", "HBM"),
    "gl_place_k": ("op_build", "bank-aware placement of a slice's entries", "LDS round trips"),
    "gl_fb_scatter_k": ("op_build", "a band's entries dealt into buckets of word positions", "HBM"),
    "gl_fb_fill_k": ("op_build", "buckets -> pass-2 stream, whole 512-byte lines assembled in LDS", "HBM"),
    "gl_fb_count_k": ("op_build", "entries per (band, bucket)", "HBM"),
    "gl_fill1_k": ("op_build", "pass-1 stream", "HBM"),
    "gl_sort2_k": ("op_build", "every (word, band) cell in ascending order (bitonic in registers)", "HBM"),
    "gl_hist_count_k": ("op_build", "(word, band) cell sizes by LDS histograms", "LDS integer atomics"),
    "gl_cnt_k": ("op_build", "super-rounds per (wave, band, group)", "HBM"),
    "gl_bst_k": ("op_build", "band boundaries per document", "HBM"),
    "csc_validate_k": ("upload", "row ids in range and ascending (once per upload)", "HBM"),
    "sumsq_k": ("frobenius", "sum of squares", "HBM"),
    "rs_scatter_k": ("sort", "radix pass", "HBM"),
    "rs_hist_k": ("sort", "radix histogram", "HBM"),
    "transpose_k": ("dense", "tiled transposition", "HBM"),
    "fetch_rows_k": ("kmpp", "seed rows of P", "launch"),
    "member_keys_k": ("kmeans", "member-list keys", "HBM"),
    "__amd_rocclr_copyBuffer": ("runtime", "hipMemcpyAsync: the expand loop's mailbox (round 6: on the copy stream beside the next application — its duration here is its wait for a free CU, off the critical path), small copies", "PCIe latency"),
    "__amd_rocclr_fillBufferAligned": ("runtime", "hipMemsetAsync", "launch"),
}


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            n = n[:i]
            break
    import re
    n = re.sub(r"^(gl_apply_k<\d, (?:true|false)), \d>", r"\1, G>", n)  # items per lane: 6 / 7 at config 3, 4 / 5 at the shard
    n = re.sub(r"^td_persist_k<true>", "td_persist_k", n)
    return n


def load(path, steps):
    out = {}
    for r in csv.DictReader(open(path)):
        k = short(r["Name"])
        out[k] = out.get(k, 0.0) + float(r["TotalDurationNs"]) / 1e6 / steps
    return out


def info(k):
    best = None
    for p in INFO:
        if k.startswith(p) and (best is None or len(p) > len(best)):
            best = p
    return INFO[best] if best else ("?", "", "")


def main():
    a, sa, b, sb = sys.argv[1], float(sys.argv[2]), sys.argv[3], float(sys.argv[4])
    N = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    A, Bm = load(a, sa), load(b, sb)
    top = sorted(A, key=lambda k: -A[k])[:N]
    for k in sorted(Bm, key=lambda k: -Bm[k])[:12]:  # the shard's own heaviest kernels, where they are not among config 3's
        if k not in top:
            top.append(k)
    print("| kernel | family | computes | bound by | ms / step, config 3 | ms / step, C3 shard |")
    print("|---|---|---|---|---|---|")
    for k in top:
        f, w, bd = info(k)
        name = k.replace("isle_gemm3::", "").replace("isle_gemm::", "").replace("|", "\\|")
        print("| `%s` | %s | %s | %s | %s | %s |" % (name[:70], f, w, bd, ("%.1f" % A[k]) if k in A else "—", ("%.2f" % Bm[k]) if k in Bm else "—"))
    print()
    print("Sum over all kernels: config 3 %.0f ms per step, C3 shard %.0f ms per step." % (sum(A.values()), sum(Bm.values())))


if __name__ == "__main__":
    main()
