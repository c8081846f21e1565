/*
 * ltr_gpu.h -- C-ABI of the MI355X (gfx950) read-vs-haplotype alignment library.
 *
 * This is the drop-in boundary for ONE path of gymrek-lab/LongTR: the batch
 * driver HapAligner::process_reads and the pair-HMM DP it runs for every
 * (pooled read, candidate haplotype) pair, plus the two thin consumers either
 * side of it.  Every entry point cites the reference interface it replaces
 * (paths relative to the reference repository root).
 *
 *   reference call site    src/seq_stutter_genotyper.cpp:517-523
 *   reference callee       src/SeqAlignment/HapAligner.h:137-138 (process_reads)
 *                          src/SeqAlignment/HapAligner.cpp:236-343 (align_seq_to_hap)
 *
 * Plain pointers and sizes only; no C++/torch types.  The library never
 * exits the process (the reference does, src/error.cpp:6-10): every function
 * returns an ltr_status and ltr_last_error() holds the text.
 *
 * Threads.  The reference is single-threaded and its HapAligner is not re-entrant (it advances the Haplotype's
 * iterator, HapAligner.cpp:852-853).  Here any number of host threads may use ONE context: plan creation is
 * serialised inside the library (the context's work arrays); ltr_calc_hap_aln_probs and ltr_haplotype_align_to_ref,
 * which stage in buffers the context keeps, run one at a time per context (a second caller waits); executes of
 * DIFFERENT plans may be queued side by side; ONE plan is driven by one thread at a time.  ltr_ctx_set_params /
 * ltr_ctx_set_pair_packing / ltr_ctx_set_debug while another thread is inside a call on the same context is the
 * caller's race.  Threads that should score at the same time take a context each.
 *
 * There is NO CPU fallback behind these entry points: if no HIP device (or no
 * gfx950 code object) is available the compute calls fail with
 * LTR_ERR_NO_DEVICE.
 */
#ifndef LTR_GPU_H_
#define LTR_GPU_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------- */
#define LTR_OK              0
#define LTR_ERR_INVALID    -1   /* bad argument / malformed batch                       */
#define LTR_ERR_NO_DEVICE  -2   /* no usable HIP device: the product path has no CPU fallback */
#define LTR_ERR_HIP        -3   /* a HIP runtime call failed                            */
#define LTR_ERR_NOMEM      -4
#define LTR_ERR_CIGAR      -5   /* CIGAR op outside MIDNSHP=X handled by trim_alignment
                                   (reference: printErrorAndDie, HapAligner.cpp:375)     */
#define LTR_ERR_UNSUPPORTED -6  /* e.g. a short-path locus that is not [flank][repeat][flank] */

/* ---- sentinels the reference writes as values (reproduced, never errors) -- */
#define LTR_IMPOSSIBLE     (-1000000000.0) /* HapAligner.cpp:20, haplotype <= 60 bp (:241-244)   */
#define LTR_ABORT_SCORE    (-700.0)        /* |n-m| > 600 (:249-252) and row abort (:300-306)    */

typedef struct ltr_ctx  ltr_ctx;    /* one per GPU; owns device memory, streams, model tables */
typedef struct ltr_plan ltr_plan;   /* one packed + uploaded batch, resident in HBM            */

/*
 * Alignment model.  Replaces class AlignmentModel (HapAligner.h:12-37) and the
 * two ints HapAligner carries (HapAligner.h:50-51).  The seven transitions stay
 * FLOAT on purpose: the reference stores them as float and promotes to double
 * inside the recurrence, and bit-exact parity depends on reproducing that.
 * Order = --alignment-params order (hipstr_main.cpp:419-439).
 */
typedef struct ltr_align_params {
  float   log_ins_to_ins;      /* a */
  float   log_ins_to_match;    /* b */
  float   log_del_to_del;      /* c */
  float   log_del_to_match;    /* d */
  float   log_match_to_match;  /* e */
  float   log_match_to_ins;    /* f */
  float   log_match_to_del;    /* g */
  int32_t indel_flank_len;     /* INDEL_FLANK_LEN, CLI default 5 (hipstr_main.cpp:324-326) */
  int32_t use_short_path;      /* SWITCH_OLD_ALIGN_LEN used as a bool (HapAligner.cpp:552) */
} ltr_align_params;

/* Fills the defaults of HapAligner.h:118 and indel_flank_len=5, use_short_path=0. */
void ltr_default_params(ltr_align_params* p);

/*
 * A batch of loci, flattened.  One locus = the inputs of one
 * HapAligner::process_reads call: P pooled, TRIMMED reads (the output of
 * HapAligner::trim_alignment, :346-465) and H candidate haplotype strings in
 * Haplotype::next() order (Haplotype.cpp:151-196), each the FULL haplotype
 * (35-bp flanks included; the library cuts hap[30:len-30] itself like :245-246).
 *
 * Output layout: for locus l the P_l x H_l log-likelihood matrix is row-major
 * at out_ll + ll_off[l], ll_off[l] = sum_{k<l} P_k*H_k  -- i.e. exactly
 * aln_probs[(pool)*H + hap] of HapAligner.cpp:550 per locus, loci back to back.
 */
typedef struct ltr_locus_batch {
  int64_t        n_loci;
  const int64_t* locus_read_off;  /* [n_loci+1] first read index of each locus            */
  const int64_t* locus_hap_off;   /* [n_loci+1] first haplotype index of each locus       */
  int64_t        n_reads;         /* = locus_read_off[n_loci]                             */
  const uint8_t* read_bytes;      /* concatenated trimmed read sequences                  */
  const int64_t* read_off;        /* [n_reads+1] byte offsets into read_bytes             */
  int64_t        n_haps;          /* = locus_hap_off[n_loci]                              */
  const uint8_t* hap_bytes;       /* concatenated full haplotype sequences                */
  const int64_t* hap_off;         /* [n_haps+1] byte offsets into hap_bytes               */
  const uint8_t* realign_read;    /* optional [n_reads]; 0 => row left untouched
                                     (HapAligner.cpp:557-560). NULL = all 1.              */
  const uint8_t* realign_hap;     /* optional [n_haps]; 0 => column left untouched
                                     (HapAligner.cpp:841-845). NULL = all 1.              */
} ltr_locus_batch;

/* ---- context ------------------------------------------------------------- */
/* device_ordinal: HIP device index (one process per GPU: pass LOCAL_RANK). */
int  ltr_ctx_create(int device_ordinal, ltr_ctx** ctx);
void ltr_ctx_destroy(ltr_ctx* ctx);
/* Replaces the alignment_model_params_ argument of the HapAligner constructor
 * (HapAligner.h:94-120).  Rebuilds the device-side boundary tables. */
int  ltr_ctx_set_params(ltr_ctx* ctx, const ltr_align_params* p);
const char* ltr_last_error(const ltr_ctx* ctx);
/* "gfx950" etc. of the device the context is bound to; number of CUs; shader clock in MHz. */
int  ltr_ctx_device_info(const ltr_ctx* ctx, char* arch, int arch_len, int* n_cu, int* clock_mhz);

/* ---- one-shot: host buffers in, host buffers out -------------------------- */
/*
 * Replaces the compute of HapAligner::process_reads (HapAligner.cpp:545-581,
 * long path) for a whole batch of loci.  out_ll: sum_l P_l*H_l doubles (cells
 * whose row/column mask is 0 are NOT written, like the reference).  out_seed:
 * n_reads ints, seed_positions[] of :562-563 (= trimmed length - 1 ... see
 * note in DESIGN.md; written only for realigned reads).  May be NULL.
 */
int ltr_align_batch(ltr_ctx* ctx, const ltr_locus_batch* batch,
                    double* out_ll, int32_t* out_seed);

/* ---- resident plan: pack + upload once, execute many times ---------------- */
int     ltr_plan_create(ltr_ctx* ctx, const ltr_locus_batch* batch, ltr_plan** plan);
void    ltr_plan_destroy(ltr_plan* plan);
int64_t ltr_plan_num_pairs(const ltr_plan* plan);   /* pairs with both masks set            */
int64_t ltr_plan_ll_size(const ltr_plan* plan);     /* number of doubles in the LL buffer   */
double  ltr_plan_cells(const ltr_plan* plan);       /* nominal DP cells, n*m per non-shortcut pair
                                                       (BASELINE.md definition)             */
double  ltr_plan_input_bytes(const ltr_plan* plan); /* algorithmic HBM bytes of one execute:
                                                       sum m + sum n + 8*P*H per locus      */
/*
 * Launch the alignment kernels for the whole plan on `stream` (a hipStream_t
 * passed as void*; NULL = the context's own stream).  Asynchronous.  d_out_ll:
 * device pointer to ltr_plan_ll_size() doubles, or NULL to use the plan's own
 * device buffer.  Inputs are already resident in HBM.
 */
int ltr_plan_execute(ltr_plan* plan, double* d_out_ll, void* stream);
/* Waits for the last execute and copies results to host (either may be NULL). */
int ltr_plan_fetch(ltr_plan* plan, double* out_ll, int32_t* out_seed);
/* Device time of the DP kernels of the last execute, measured with HIP events
 * recorded on the launch stream; n_launches = kernel launches it took. */
int ltr_plan_last_kernel_ms(ltr_plan* plan, float* ms, int* n_launches);
/* The DP runs as one launch per strip-width class (k = 0 .. ltr_num_kernels()-1, read
 * columns per lane = *strip_width; 0 = chosen per pair).  Per class: pairs, nominal cells, and the device time of
 * its launch in the last execute (HIP events on the launch stream). */
int ltr_num_kernels(void);
/* Lanes that share one pair in class k: 64 (one pair per wavefront), 2 .. 32 (packed classes: 64 / lanes
 * short reads per wavefront, each on a segment of the lanes with wider strips), 256 / 512 (a workgroup per pair). */
int ltr_kernel_lanes_per_pair(int k);
/* Kernel family of class k: 0 one pair per wavefront, 1 several pairs per wavefront (packed), 2 one pair per workgroup
 * (certificate kernels); 3 exact (redo) kernels -- the classes at the end of the list: pairs a certificate
 * could not clear, by read length, plus the byte-compare kernel for pairs with bytes outside ACGT.  For
 * them ltr_plan_kernel_stats reports the pairs they scored in the last execute (cells: only of the
 * pairs that start out in their lists). */
int ltr_kernel_family(int k);
/* Per-launch HIP events cost a few microseconds each: off by default, switch on before the
 * executes whose ltr_plan_kernel_stats times you want. */
int ltr_plan_set_timing(ltr_plan* plan, int on);
int ltr_plan_kernel_stats(ltr_plan* plan, int k, int* strip_width, int64_t* n_pairs, double* cells, float* ms);
/* Launches that score several classes: the several-pairs-per-wavefront kernels run ONE launch per strip width over up to five
 * ranges of pairs (32, 16, 8, 4, 2 lanes per pair), and in automatic mode the one-pair-per-wavefront classes of strip widths
 * 11 .. 20 are ONE persistent launch that walks them widest first.  ltr_plan_kernel_stats reports such a launch under its first
 * class that has pairs, the others report zero.  Returns the number of ranges of class k's launch (0: k is not such a class)
 * and fills lanes_per_pair / strip_width / n_pairs (room for ltr_num_kernels() each) in launch order.  (In automatic mode the
 * packed launches of strip widths 13 .. 20 are one persistent launch as well; and every plan of the automatic mode, whatever its
 * size and indel model, runs EVERY one-wave class and packed width as one launch, the plan kernel -- csrc/ltr_dp_plan.hpp -- which
 * also scores the pairs whose certificate fails itself: its ranges are the one-wave classes, then the packed widths.)
 * ltr_plan_set_timing: on = 1 times every launch as it is launched; on = 2 launches the multi-width launch class by class
 * (the single-class kernels: the same bodies) so that every class has a time of its own. */
int ltr_plan_kernel_ranges(ltr_plan* plan, int k, int32_t* lanes_per_pair, int32_t* strip_width, int64_t* n_pairs);
/* The class the plan kernel's launch is reported under by ltr_plan_kernel_stats / _ranges (-1: this plan runs a launch per class). */
int ltr_plan_kernel_class(const ltr_plan* plan);
/* Measurement aid: with ltr_ctx_set_debug(ctx, "wave_clock", 1) set when the plan was created, the plan kernel records the wall
 * clock (100 MHz) at which every one of its wavefronts started and left, and what it spent in the exact body: out = {first, last,
 * pairs scored with the exact body, ticks spent there} per wavefront of the last execute, then 4096 words: a count and
 * (n << 32 | m) of the pairs that took the exact body, then 256 words: per entry of the plan kernel's table (ltr_plan_debug_entries) the
 * ticks all wavefronts together spent in it.  cap (64-bit words) must hold 4 x wavefronts + 4096 + 256.
 * Returns the number of wavefronts written (0: not recorded), negative on error. */
int ltr_plan_debug_wave_clocks(ltr_plan* plan, uint64_t* out, int64_t cap);
/* The plan kernel's table, in walk order: kind (0 one pair per wavefront, 1 packed strip width, 2 pairs that start with the exact
 * body, 3 = 0 by the chained walk), strip width, pairs, nominal cells of every entry.  Returns the number of entries. */
int ltr_plan_debug_entries(const ltr_plan* plan, int32_t* kind, int32_t* strip_width, int64_t* n_pairs, double* cells, int cap);

/* ---- planning units (host only, no GPU needed): which kernel scores a pair, in what order ------------------
 * ltr_plan_create = validate -> one launch class + launch-order key per pair -> counting sort by class, longest
 * first inside a class -> upload.  The rule and the sort are plain host functions (csrc/ltr_plan.cpp); these entry
 * points expose them to the CPU tests.  They decide scheduling only: every kernel returns the reference's bits
 * (HapAligner.cpp:236-343).
 *   ltr_debug_class_info   family (ltr_kernel_family), strip width, waves per pair, lanes per pair of class k
 *   ltr_debug_classify     the class / order key / exact list of ONE pair of a batch with `pairs_in_batch` pairs
 *                          (`long_pairs_in_batch` of them with reads over 1281 bases) on a GPU of n_cu CUs under
 *                          ltr_ctx_set_pair_packing mode `mode`; window_len = haplotype window (0 when
 *                          hap_full_len <= 60), generic = bytes outside ACGT
 *   ltr_debug_sort_by_class  order[i] = pair at sorted position i; class_first[k] .. class_first[k+1] = class k;
 *                          fold: 0 = none; 1 = under-filled classes are folded into the next wider one (automatic mode, a
 *                          launch per class); 2 = the same where the widths 11.. / 13.. share a multi-width launch (nothing is
 *                          folded inside those); 3 = under the plan kernel (one launch for every one-wave and packed class:
 *                          only the workgroup families fold) */
int ltr_debug_num_classes(void);
int ltr_debug_class_info(int k, int* family, int* strip_width, int* waves_per_pair, int* lanes_per_pair);
int ltr_debug_classify(const ltr_align_params* p, int mode, int n_cu, int64_t pairs_in_batch, int64_t long_pairs_in_batch,
                       int32_t window_len, int32_t read_len, int32_t hap_full_len, int generic,
                       int* launch_class, int* order_key, int* exact_list);
int ltr_debug_sort_by_class(const int16_t* launch_class, const int16_t* order_key, int64_t n_pairs, int fold, int n_cu,
                            int32_t* order, int32_t* class_first);
/* The modelled launch time (wave-cycles: steps x (strip width + per-step overhead) x the share of the wavefront the pair holds) the
 * launch-order key is the logarithm of, for n pairs at once: what a host shards a catalogue by (longtr_amd/shard.py). */
int ltr_debug_pair_costs(const ltr_align_params* p, int mode, int n_cu, int64_t pairs_in_batch, int64_t long_pairs_in_batch, int64_t n,
                         const int32_t* window_len, const int32_t* read_len, const int32_t* hap_full_len, double* cost);
/* The exact kernels' row test as a table: entry k + H (H = half the returned length) = the smallest double x with
 * fl(x + (double)((float)|k| * log_del_to_del)) >= -600, +inf where no negative cell value can pass (the reference's abort test,
 * HapAligner.cpp:297-306, only compares the row maximum with -600).  Returns the table's length (cap must be >= 2110). */
int ltr_debug_threshold_table(float log_del_to_del, double* out, int64_t cap);
/* The seeded path's host-side seed choice = HapAligner::calc_seed_base + calc_best_seed_position (HapAligner.cpp:467-542):
 * index of the seed base in the read, -1 none, -2 a CIGAR op the reference dies on (forward declarations: types below). */
struct ltr_alignment;
struct ltr_haplotype_blocks;
int ltr_debug_calc_seed_base(const struct ltr_alignment* aln, const struct ltr_haplotype_blocks* hap);
/* The host-thread budget (see "host threads" below) as the CPU tests see it: the number of distinct threads a parallel loop of
 * n_items really ran on under a budget of n_threads (0: the rule) in worker pool which_pool (0 / 1), and whether
 * ltr_calc_hap_aln_probs stages its next chunk on a helper thread under that budget (1 / 0). */
int ltr_debug_parallel_threads(int n_threads, int64_t n_items, int which_pool);
int ltr_debug_prep_ahead_rule(int n_threads);

/* ---- host-side mirror of the reference objects (flattened) ---------------- */
/*
 * One haplotype = an ordered list of blocks (Haplotype.h:12-16); block b has
 * n_alleles[b] alternative sequences (HapBlock.h:19-21; allele 0 = reference)
 * and reference coordinates [start,end).  is_repeat marks RepeatBlock
 * (RepeatBlock.h:15).  Sequences of all alleles of all blocks are concatenated
 * in (block, allele) order.
 */
typedef struct ltr_haplotype_blocks {
  int32_t        n_blocks;
  const int32_t* block_start;     /* [n_blocks] HapBlock::start()                */
  const int32_t* block_end;       /* [n_blocks] HapBlock::end()                  */
  const uint8_t* is_repeat;       /* [n_blocks] get_repeat_info() != NULL        */
  const int32_t* period;          /* [n_blocks] repeat period (0 for flanks)     */
  const int32_t* n_alleles;       /* [n_blocks] HapBlock::num_options()          */
  const uint8_t* allele_bytes;    /* concatenated allele sequences               */
  const int64_t* allele_off;      /* [sum n_alleles + 1]                         */
} ltr_haplotype_blocks;

/* class Alignment (AlignmentData.h:28-140): the fields the path reads. */
typedef struct ltr_alignment {
  int32_t        start;           /* get_start()  */
  int32_t        stop;            /* get_stop()   */
  const uint8_t* seq;             /* get_sequence() */
  int32_t        seq_len;
  int32_t        n_cigar;
  const char*    cigar_type;      /* [n_cigar] CigarElement::get_type() */
  const int32_t* cigar_num;       /* [n_cigar] CigarElement::get_num()  */
  const uint8_t* qual;            /* get_base_qualities(), Phred+33, seq_len bytes; only the short
                                     (stutter) path reads it -- may be NULL on the long path */
} ltr_alignment;

/* Stutter model of the short path.  Replaces StutterModel (stutter_model.h:35-62); the CLI always
 * installs (0.95, 0.05, 0.05, 0.95, 0.01, 0.01) (hipstr_main.cpp:140,362-363). */
typedef struct ltr_stutter_params {
  double in_geom, in_up, in_down;      /* in-frame geometric parameter, P(up), P(down)     */
  double out_geom, out_up, out_down;   /* out-of-frame                                      */
} ltr_stutter_params;
void ltr_default_stutter_params(ltr_stutter_params* p);
int  ltr_ctx_set_stutter_params(ltr_ctx* ctx, const ltr_stutter_params* p);
/* Scheduling knob, results never depend on it.  Reads of up to 641 bases can share a wavefront
 * with other pairs (segments of 32 .. 2 lanes each): better throughput, longer latency per pair.  Reads longer
 * than 1281 bases are scored by a whole workgroup (4 or 8 wavefronts, boundary columns handed
 * over through LDS) while there are fewer than 10 such pairs per CU, else as several column
 * blocks on one wavefront each; reads of 3586 .. 5121 bases by four-wave workgroups with wide strips up to 80
 * long pairs per CU, for as many whole rounds of workgroups as the batch fills.
 *   -1 (default) automatic: several pairs per wavefront from 32 pairs per CU up, the fewer lanes per pair the more
 *                pairs of that length the batch holds; launch classes that cannot fill the GPU a few times over
 *                are folded into the next wider class; from 16 pairs per CU up the launches of a plan are dealt
 *                over four streams of the context
 *    0 / 1       packed kernels never / 32 lanes per pair whenever the read fits (one class per strip width, one stream)
 *    2           workgroup kernels wherever they exist: reads over 1025 bases, and the one-wave variant for every
 *                read of up to 1025 bases (inputs streamed through LDS; A/B and tests -- slower than the default)
 *    3           no workgroup kernels at all (long reads walk their column blocks on one wavefront)
 *    4           no certificate kernels: every pair goes straight to the exact kernels (the reference's
 *                cell-by-cell row maximum) -- verification, and the rate of the exact kernels by themselves
 *    5 .. 8      packed kernels with 16 / 8 / 4 / 2 lanes per pair whenever the read fits that many lanes of up
 *                to 20 columns (else the next wider segment; beyond 641 bases one pair per wavefront)
 * Workgroup kernels exist for symmetric indel models (ins->match == del->match, match->ins ==
 * match->del: the defaults); a plan that uses them must be re-created if ltr_ctx_set_params switches
 * to an asymmetric model (ltr_plan_execute reports LTR_ERR_INVALID otherwise). */
int  ltr_ctx_set_pair_packing(ltr_ctx* ctx, int mode);
/* Measurement switches (A/B runs, profile collection); results never depend on them; value 0 = the library's rule.
 *   "fan_lanes"      streams the certificate launches of a plan are dealt over (1 .. 4; rule: 2) -- profiles/collect.sh
 *                    sets 1 so that every launch's own duration is what rocprofv3 and the per-launch HIP events see
 *   "fan_pairs"      plans with at least this many pairs stay on one stream
 *   "chunks", "chunk_streams", "chunk_growth"   ltr_calc_hap_aln_probs: chunk count / streams / size progression
 *   "trace"          1: ltr_calc_hap_aln_probs prints a timestamped phase profile to stderr
 *   "wg_first_pass"  first pass of the workgroup-per-pair classes (long reads): 1 = always the certificate kernels, 2 = always the
 *                    threshold kernels; rule: what the context has learnt, see ltr_ctx_wg_first_pass
 *   "reset"          every switch back to the library's rule */
int  ltr_ctx_set_debug(ltr_ctx* ctx, const char* key, double value);
/* Which kernels score the workgroup-per-pair classes (reads beyond 1281 bases, symmetric models) FIRST in the next
 * ltr_plan_execute: 0 = the certificate kernels (11 operations a cell; a pair whose certificate fails -- it would abort in the
 * reference, HapAligner.cpp:297-306, or comes close -- is scored again by an exact kernel), 1 = the threshold kernels (13
 * operations a cell, exact in one pass).  The context LEARNS it from the reads: every execute leaves the number of pairs its
 * first pass could not finish; more than half of them failing switches to the threshold kernels, fewer than a quarter aborting
 * switches back; ltr_ctx_set_params forgets it.  HiFi reads finish (0); ONT reads under the default model abort (1: every pair
 * of BASELINE config 5).  Scores never depend on it.  last_unfinished / last_scored (optional): the counts of the last execute
 * that has been read.  The query waits for the statistics of the executes queued so far (ltr_plan_execute never waits: it reads
 * what has arrived).  Returns 0 / 1, or a negative error. */
int  ltr_ctx_wg_first_pass(ltr_ctx* ctx, int64_t* last_unfinished, int64_t* last_scored);

/*
 * HapAligner::process_reads (HapAligner.h:137-138, .cpp:545-581) for one locus:
 * trims every alignment on the host exactly like trim_alignment (:346-465,
 * incl. the 10-bp substitute for an empty trim, :820-823), enumerates the
 * haplotypes in Haplotype::next() order and scores all pairs on the GPU.
 * aln_probs[(init_read_index+i)*H + k] and seed_positions[init_read_index+i]
 * are written for realign_read[i] != 0 and realign_to_hap[k] != 0 only.
 */
int ltr_process_reads(ltr_ctx* ctx, const ltr_haplotype_blocks* hap,
                      const uint8_t* realign_to_hap,
                      const ltr_alignment* alns, int32_t n_alns, int32_t init_read_index,
                      const uint8_t* realign_read,
                      double* aln_probs, int32_t* seed_positions);

/* Number of haplotype combinations (Haplotype::num_combs) and the k-th
 * haplotype string in Haplotype::next() order; used by callers that need to
 * flatten several loci into one ltr_locus_batch themselves. */
int64_t ltr_haplotype_num_combs(const ltr_haplotype_blocks* hap);
/* Writes haplotype `index` into out (capacity cap); returns its length or <0. */
int64_t ltr_haplotype_seq(const ltr_haplotype_blocks* hap, int64_t index, uint8_t* out, int64_t cap);
/* HapAligner::trim_alignment (:346-465): returns ltrim/rtrim (bases cut left/right). */
int ltr_trim_alignment(const ltr_alignment* aln, int32_t repeat_start, int32_t repeat_end,
                       int32_t indel_flank_len, int32_t* ltrim, int32_t* rtrim);
/* ReadPooler::add_alignment over a whole read list (read_pooler.cpp:3-20):
 * pool_index[i] = pool of read i (pools numbered by first occurrence);
 * returns the number of pools or <0. */
int32_t ltr_pool_reads(const uint8_t* const* seqs, const int32_t* seq_lens, int32_t n_reads,
                       int32_t* pool_index);
/* SeqStutterGenotyper::calc_hap_aln_probs scatter (seq_stutter_genotyper.cpp:526-559):
 * pool rows -> read rows (masked), then mate-pair row sums. */
int ltr_scatter_pool_probs(const double* log_pool_aln_probs, const int32_t* pool_seed_positions,
                           const int32_t* pool_index, int32_t n_reads, int32_t n_alleles,
                           const uint8_t* realign_to_hap, const uint8_t* copy_read,
                           const uint8_t* second_mate,
                           double* log_aln_probs, int32_t* seed_positions);

/*
 * SeqStutterGenotyper::calc_hap_aln_probs (seq_stutter_genotyper.cpp:514-563) for MANY loci in one
 * GPU pass -- the throughput form of the drop-in.  Per locus: the R reads as the genotyper holds
 * them (alns_, after left_align_reads) and the haplotype blocks.  The library pools the reads
 * (ReadPooler), trims each pool (HapAligner::trim_alignment), scores every pool x haplotype pair of
 * every locus together, and fans the rows back out to reads (+ mate-pair sums).
 * log_aln_probs[l]: R_l x H_l doubles, seed_positions[l]: R_l ints (caller-owned, like
 * log_aln_probs_ / seed_positions_ of genotyper.h:37, seq_stutter_genotyper.h).
 */
typedef struct ltr_locus {
  const ltr_haplotype_blocks* hap;
  const ltr_alignment*        alns;
  int32_t                     n_alns;
  const uint8_t*              second_mate;   /* optional [n_alns], second_mate_ (seq_stutter_genotyper.cpp:491-497) */
  /* The three masks of calc_hap_aln_probs(realign_to_haplotype, realign_pool, copy_read) -- the form
   * add_and_remove_alleles calls after new candidate haplotypes were added (seq_stutter_genotyper.cpp:390-391):
   * only flagged haplotype columns / pools are scored and only flagged reads' rows are rewritten; mate rows
   * are summed over the flagged columns only (:546-559).  Each optional, NULL = all set (the call at :634). */
  const uint8_t*              realign_to_hap; /* [H]      */
  const uint8_t*              realign_pool;   /* [P] pools in order of first occurrence (ReadPooler) */
  const uint8_t*              copy_read;      /* [n_alns] */
} ltr_locus;
int ltr_calc_hap_aln_probs(ltr_ctx* ctx, const ltr_locus* loci, int64_t n_loci,
                           double* const* log_aln_probs, int32_t* const* seed_positions);

/* ---- consumer: genotype posteriors ---------------------------------------- */
/*
 * Genotyper::calc_log_sample_posteriors + get_optimal_haplotypes
 * (genotyper.cpp:21-100) on the GPU.  log_aln_probs [R x H] is clamped to
 * >= -600 IN PLACE like genotyper.cpp:57-58.  log_sample_posteriors [S x H x H],
 * sample_total_ll [S], gts [S x 2] (argmax, first maximum in row-major order).
 * Returns total LL through *total_ll.
 */
int ltr_posteriors(ltr_ctx* ctx, int32_t n_samples, int32_t n_reads, int32_t n_alleles,
                   double* log_aln_probs, const double* log_p1, const double* log_p2,
                   const int32_t* sample_label, int32_t haploid,
                   double* log_sample_posteriors, double* sample_total_ll,
                   int32_t* gts, double* total_ll);

/*
 * The same consumer for EVERY locus of a resident plan, reading the log-likelihood buffer of
 * the last ltr_plan_execute in place on the device (what SeqStutterGenotyper::genotype does per
 * locus: calc_hap_aln_probs -> calc_log_sample_posteriors, seq_stutter_genotyper.cpp:632-634).
 * Reads map to pool rows through pool_index (the scatter of seq_stutter_genotyper.cpp:526-538
 * without materialising the per-read matrix).  Outputs, loci then samples in order:
 * log_sample_posteriors [sum_l S_l*H_l*H_l], sample_total_ll [sum_l S_l], gts [2*sum_l S_l].
 */
typedef struct ltr_posterior_batch {
  int64_t        n_loci;          /* must equal the plan's */
  const int64_t* locus_read_off;  /* [n_loci+1] first READ (not pool) of each locus            */
  int64_t        n_reads;
  const int32_t* pool_index;      /* [n_reads] pool row of the read inside its locus          */
  const double*  log_p1;          /* [n_reads] phasing log-likelihoods (snp_bam_processor.h:17-18) */
  const double*  log_p2;
  const int32_t* sample_label;    /* [n_reads] sample index inside the locus                  */
  const int32_t* n_samples;       /* [n_loci]                                                 */
  int32_t        haploid;
} ltr_posterior_batch;
int ltr_plan_posteriors(ltr_plan* plan, const ltr_posterior_batch* pb,
                        double* log_sample_posteriors, double* sample_total_ll, int32_t* gts);

/* ---- consumer, next step: genotype fields ------------------------------------ */
/*
 * Genotyper::extract_genotypes_and_likelihoods (genotyper.cpp:132-256) with calc_PLs (:102-107)
 * and calc_gl_diff (:109-130): from the normalised posterior matrix (ltr_posteriors /
 * ltr_plan_posteriors) to the per-sample values the VCF writer prints
 * (seq_stutter_genotyper.cpp:1255-1300: GT, Q = exp(log_unphased), PQ = exp(log_phased), GLDIFF,
 * GL, PL, PHASEDGL).  Host code: libm exp/log and the FP32 fastlog/fastexp of fast_log_sum_exp
 * (mathops.cpp:87-96), evaluated in the reference's order.
 *   n_alleles      number of haplotypes H (Genotyper::num_alleles_)
 *   n_variants     number of alleles V of the variant being reported
 *   hap_to_allele  [H] haplotype -> allele of that variant
 *   best_haplotypes[S x 2] get_optimal_haplotypes (the `gts` output of ltr_posteriors)
 * Every output pointer is optional (NULL = not wanted); gl_diffs, gls, pls and phased_gls are the
 * calc_gls / calc_pls / calc_phased_gls branches (:204-255).
 *   gls, pls    [S x n_gl], n_gl = haploid ? V : V(V+1)/2, order (i1, i2 <= i1) as VCF GL
 *   phased_gls  [S x (haploid ? V : V*V)]
 */
typedef struct ltr_genotype_fields {
  int32_t* best_gts;                      /* [S x 2] */
  double*  log_phased_posteriors;         /* [S] */
  double*  log_unphased_posteriors;       /* [S] */
  double*  hap_log_phased_posteriors;     /* [S] */
  double*  hap_log_unphased_posteriors;   /* [S] */
  double*  gls;
  double*  gl_diffs;                      /* [S] */
  int32_t* pls;
  double*  phased_gls;
} ltr_genotype_fields;
int ltr_extract_genotypes(int32_t n_samples, int32_t n_alleles, int32_t n_variants,
                          const int32_t* hap_to_allele, int32_t haploid,
                          const double* log_sample_posteriors, const double* sample_total_ll,
                          const int32_t* best_haplotypes, const ltr_genotype_fields* out);

/* ---- upstream of the path: raw reads -> prepared reads -> candidate haplotypes (host) ---- */
/* A BAM record as the locus driver hands it over (BamAlignment, bam_io.h): only what the path reads. */
typedef struct ltr_raw_alignment {
  int32_t        pos;             /* Position(), 0-based */
  int32_t        end_pos;         /* GetEndPosition(): first reference base after the alignment */
  const uint8_t* bases;           /* QueryBases() */
  const uint8_t* quals;           /* Qualities(), Phred+33; may be NULL */
  int32_t        length;
  int32_t        n_cigar;
  const char*    cigar_type;      /* CigarData(): M = X I D S H */
  const int32_t* cigar_num;
  int32_t        sample;          /* index of the read's sample (read group) */
  int32_t        haplotype_tag;   /* HP tag: 1 / 2, 0 = none (genotyper_bam_processor.cpp:145-150) */
  uint8_t        reverse;         /* IsReverseStrand() */
  uint8_t        use_for_hap_generation;   /* the read's "PF" flag for this region (bam_processor.cpp:28-35) */
} ltr_raw_alignment;
typedef struct ltr_read_set ltr_read_set;       /* the locus' prepared reads (left_alns), owned by the library */
/*
 * GenotyperBamProcessor::left_align_reads (genotyper_bam_processor.cpp:38-168) for one locus: reads that do not
 * span the region are dropped (:56), the rest are cut to region -+ 200 bp (BamAlignment::TrimAlignment,
 * bam_io.cpp:267-372), reads with the repeat deleted become "deleted" placeholders (:62-71), M/=/X runs are
 * re-derived against the reference (:80-135), soft-clipped reads are dropped (:137-140), HP tags are counted.
 * chrom_seq is a window of the chromosome that starts at coordinate chrom_seq_start.
 */
int ltr_left_align_reads(const ltr_raw_alignment* raw, int32_t n_raw, int32_t n_samples, int32_t region_start, int32_t region_stop,
                         const uint8_t* chrom_seq, int64_t chrom_seq_start, int64_t chrom_seq_len, ltr_read_set** out);
/*
 * SNPBamProcessor::process_phased_reads (snp_bam_processor.cpp:141-226, --phased-bam) for unpaired reads: the phasing priors
 * log_p1 / log_p2 that Genotyper::calc_log_sample_posteriors weighs a read's two haplotypes with, from the reads' HP tags.
 * haplotype[r] = 1 / 2 (the tag) or -1 (no tag, get_haplotype :126-134); read groups are visited in order 0 .. n_samples-1
 * with the reference's running totals: once a group has more than 20 % untagged reads or at most one read of either
 * haplotype (counted over the groups so far), no read of that and any later group is phased.  phased_reads may be NULL.
 */
int ltr_phasing_priors(int32_t n_reads, const int32_t* sample_of_read, const int32_t* haplotype, int32_t n_samples,
                       double* log_p1, double* log_p2, int32_t* phased_reads);
int32_t              ltr_read_set_size(const ltr_read_set* rs);
const ltr_alignment* ltr_read_set_alignments(const ltr_read_set* rs);          /* what ltr_calc_hap_aln_probs takes */
const char* const*   ltr_read_set_alignment_strings(const ltr_read_set* rs);   /* Alignment::get_alignment(): bases, '-' for deleted reference bases */
const uint8_t*       ltr_read_set_deleted(const ltr_read_set* rs);             /* Alignment::get_deleted() */
const int32_t*       ltr_read_set_source(const ltr_read_set* rs);              /* index of the raw read */
const int32_t*       ltr_read_set_sample(const ltr_read_set* rs);
const int32_t*       ltr_read_set_n_p1s(const ltr_read_set* rs);               /* [n_samples] reads tagged HP:1 / HP:2 */
const int32_t*       ltr_read_set_n_p2s(const ltr_read_set* rs);
int32_t              ltr_read_set_fail_count(const ltr_read_set* rs);          /* align_fail_count */
void                 ltr_read_set_free(ltr_read_set* rs);
/* HaplotypeGenerator::extract_sequence (HaplotypeGenerator.cpp:84-165) for read i: the read's bases between two
 * reference coordinates.  Returns their number (>= 0; out may be NULL), -1 if the read does not span, < -1 on error. */
int64_t ltr_extract_sequence(const ltr_read_set* rs, int32_t i, int32_t region_start, int32_t region_end, uint8_t* out, int64_t cap);
/*
 * SeqStutterGenotyper::build_haplotype (seq_stutter_genotyper.cpp:416-482) for one region with alleles taken from
 * the reads: HaplotypeGenerator::add_haplotype_block (:530-578: region padded by indel_flank_len, candidate
 * alleles by the exact-sequence rules of gen_candidate_seqs :295-373, sorted and trimmed :474-480, :14-82) and
 * fuse_haplotype_blocks (:580-607) -> [flank <= 35 bp][repeat block][flank <= 35 bp].  The partial-order-alignment
 * clustering of reads without a candidate (:376-472, spoa) is NOT performed; the result says how many reads /
 * samples the reference would have clustered.  ctx may be NULL (it only receives the hap-build time).
 * A failed construction (the reference's failure_msg_) is not an error status: blocks() is NULL, failure() the text.
 */
typedef struct ltr_hap_result ltr_hap_result;
int ltr_build_haplotype(ltr_ctx* ctx, const ltr_read_set* rs, int32_t n_samples, int32_t region_start, int32_t region_stop, int32_t period,
                        const uint8_t* chrom_seq, int64_t chrom_seq_start, int64_t chrom_seq_len, int64_t chrom_len,
                        int32_t indel_flank_len, ltr_hap_result** out);
const ltr_haplotype_blocks* ltr_hap_result_blocks(const ltr_hap_result* r);
const char* ltr_hap_result_failure(const ltr_hap_result* r);
int32_t     ltr_hap_result_unplaced_reads(const ltr_hap_result* r);
int32_t     ltr_hap_result_samples_needing_clustering(const ltr_hap_result* r);
void        ltr_hap_result_free(ltr_hap_result* r);

/* ---- neighbour of the path: haplotype -> reference-haplotype alignment (GPU) -------- */
/*
 * Haplotype::aln_haps_to_ref (src/SeqAlignment/Haplotype.cpp:58-86) for every haplotype of every locus in one
 * launch: NeedlemanWunsch::Align(reference haplotype, haplotype, use_ref_end_penalty = true)
 * (NeedlemanWunsch.cpp:380-420), Haplotype::adjust_indels (Haplotype.cpp:8-56) and the M / I / D strings
 * hap_aln_info_ holds.  Haplotypes need three blocks (adjust_indels asserts it).  aln_info receives the
 * strings back to back; haplotype k of locus l (Haplotype::next() order, the reference haplotype first) is
 * [info_off[h], info_off[h+1]) with h = (haplotypes of the loci before l) + k; info_off has sum_l H_l + 1
 * entries; cap >= ltr_haplotype_aln_info_capacity().  In the reference only the traced short path reads
 * these strings (HapAligner.cpp:969); the long-read path does not need this call at all.
 */
int64_t ltr_haplotype_aln_info_capacity(const ltr_haplotype_blocks* const* haps, int64_t n_loci);
int ltr_haplotype_align_to_ref(ltr_ctx* ctx, const ltr_haplotype_blocks* const* haps, int64_t n_loci,
                               char* aln_info, int64_t cap, int64_t* info_off);

/* ---- consumer, last steps: allele pruning and the VCF record (host) -------------- */
/* SeqStutterGenotyper::haps_to_alleles (seq_stutter_genotyper.cpp:240-248): allele of `block` in every haplotype. */
int ltr_haps_to_alleles(const ltr_haplotype_blocks* hap, int32_t block, int32_t* hap_to_allele);
/*
 * SeqStutterGenotyper::get_unused_alleles(check_spanned = false, check_called = true) (:250-308) for one block:
 * non-reference alleles that no called sample with an aligned read carries in its optimal haplotype pair
 * (best_haplotypes [S x 2] = the gts of ltr_posteriors; sample_filtered[s] != 0 <=> call_sample_[s] not empty).
 * Writes the allele indices to `unused` and returns their number (genotype() then calls remove_alleles, :636-645).
 */
int32_t ltr_unused_alleles(int32_t n_samples, const int32_t* best_haplotypes, const uint8_t* sample_has_aligned_read,
                           const uint8_t* sample_filtered, int32_t n_haplotypes, const int32_t* hap_to_allele,
                           int32_t n_block_alleles, int32_t* unused);
/*
 * The bookkeeping of add_and_remove_alleles (:317-409) around the re-alignment.  The caller rebuilds its block
 * list (HapBlock::remove_alleles / add_alternate); haplotypes of the old and the new list are matched by
 * sequence: allele_mapping[old haplotype] = new index or -1, realign_to_hap[new haplotype] = 1 for sequences that
 * did not exist before (the mask of the three-argument calc_hap_aln_probs, ltr_locus.realign_to_hap);
 * ltr_remap_aln_probs moves the surviving columns of log_aln_probs_ (everything else -100000, :367).
 */
int ltr_remap_haplotypes(const ltr_haplotype_blocks* old_hap, const ltr_haplotype_blocks* new_hap,
                         int32_t* allele_mapping, uint8_t* realign_to_hap);
int ltr_remap_aln_probs(const double* old_ll, int32_t n_reads, int32_t h_old, const int32_t* allele_mapping, int32_t h_new, double* new_ll);

/* Genotyper's output switches (genotyper.cpp:339-346; CLI --output-gls etc., hipstr_main.cpp:178-183). */
typedef struct ltr_vcf_options {
  int32_t output_gls, output_pls, output_phased_gls, output_allreads, output_mallreads, output_filters, output_haplotype_data;
  float   max_flank_indel_frac;
} ltr_vcf_options;
void ltr_default_vcf_options(ltr_vcf_options* o);

/* Everything SeqStutterGenotyper::write_vcf_record (:894-1366) reads for one repeat block of one locus. */
typedef struct ltr_vcf_locus {
  const char*    chrom;                 /* Region::chrom() */
  int32_t        region_start, region_stop;   /* Region::start() / stop() (0-based, half open) */
  const char*    name;                  /* Region::name(), NULL or "" -> "." */
  const char*    motif;                 /* Region::motif() */
  const char*    period_str;            /* Region::period_str() */
  const uint8_t* chrom_seq;             /* reference bases from coordinate chrom_seq_start on (a window is enough: */
  int64_t        chrom_seq_start;       /* get_alleles reads [min(region_start, block start) - 1, region_stop))    */
  int64_t        chrom_seq_len;
  const ltr_haplotype_blocks* hap;
  int32_t        block;                 /* hap_block_index: the repeat block the record is for */
  const uint8_t* inexact_allele;        /* optional [alleles of the block] HapBlock::get_inexact */
  int32_t        n_reads, n_samples, haploid;
  const double*  log_aln_probs;         /* [R x H] log_aln_probs_ (ltr_calc_hap_aln_probs) */
  const double*  log_p1; const double* log_p2;   /* [R] */
  const int32_t* sample_label;          /* [R] */
  const ltr_alignment* alns;            /* optional [R]: ALLREADS (ExtractCigar over region +- 5) */
  const uint8_t* aln_deleted;           /* optional [R] Alignment::get_deleted() */
  const double*  log_sample_posteriors; /* [S x H x H] (ltr_posteriors) */
  const double*  sample_total_ll;       /* [S] */
  const int32_t* best_haplotypes;       /* [S x 2] */
  const int32_t* n_p1s; const int32_t* n_p2s;    /* optional [S] reads per phased haplotype (PDP) */
  const char* const* sample_names;      /* [S] the genotyper's samples */
  const char* const* sample_filter;     /* optional [S] call_sample_: NULL / "" = called, else the filter reason */
  int32_t        n_out_samples;         /* VCF columns; 0 = the genotyper's samples in order */
  const char* const* out_sample_names;
} ltr_vcf_locus;
/* get_alleles (:688-785): number of alleles; text in `out`, allele i at [allele_off[i], allele_off[i+1]); *pos 1-based. */
int32_t ltr_get_alleles(const ltr_vcf_locus* v, int32_t* pos, char* out, int64_t cap, int64_t* allele_off);
/* write_vcf_record for the long-read path (no alignment traces: DFLANKINDEL = 0): the VCF line without the
 * trailing newline, NUL-terminated.  Returns its length or a negative status. */
int64_t ltr_vcf_record(const ltr_vcf_locus* v, const ltr_vcf_options* opt, char* out, int64_t cap, int32_t* pos);
/* Genotyper::get_vcf_header (genotyper.cpp:258-336): ##fileformat, ##command, ##reference, the FASTA's ##contig lines
 * (contig_lines = the text ltr_fasta_contig_lines gives, or NULL), the ##INFO / ##FORMAT definitions of every field
 * ltr_vcf_record writes (the optional ones by *opt, NULL = defaults), the #CHROM line with the sample names.
 * NUL-terminated text; returns its length or a negative status. */
int64_t ltr_vcf_header(const char* fasta_path, const char* full_command, const char* contig_lines, const ltr_vcf_options* opt,
                       const char* const* sample_names, int32_t n_samples, char* out, int64_t cap);

/* ---- on-disk formats that need no htslib ------------------------ */
/*
 * Where the reference ends the process on a malformed input (printErrorAndDie, src/error.cpp:6-10) these entry
 * points return LTR_ERR_INVALID and copy the reference's message into err (NUL-terminated, at most err_cap bytes;
 * err may be NULL).  BAM / CRAM input stays with the host program.
 */
/* readRegions (src/region.cpp:26-65): whitespace-separated CHROM START STOP MOTIF [NAME], START 1-based (stored
 * 0-based), at most max_regions regions, optionally only those on chrom_limit (NULL / "" = all).  orderRegions
 * (:67-69) = ltr_region_set_order.  period = Region::computePeriod (region.h:32-39: the common motif length, -1 when
 * the comma-separated motifs differ in length), period_str = Region::period_str (:63-71). */
typedef struct ltr_region_set ltr_region_set;
int         ltr_read_regions(const char* path, uint32_t max_regions, const char* chrom_limit, ltr_region_set** out, char* err, int err_cap);
int64_t     ltr_region_set_size(const ltr_region_set* rs);
int32_t     ltr_region_set_lines_read(const ltr_region_set* rs);     /* the reference's "Region file contains N regions" */
void        ltr_region_set_order(ltr_region_set* rs);
void        ltr_region_set_free(ltr_region_set* rs);
const char* ltr_region_chrom(const ltr_region_set* rs, int64_t i);
const char* ltr_region_name(const ltr_region_set* rs, int64_t i);     /* "" when the line has no fifth column */
const char* ltr_region_motif(const ltr_region_set* rs, int64_t i);
const char* ltr_region_period_str(const ltr_region_set* rs, int64_t i);
int32_t     ltr_region_start(const ltr_region_set* rs, int64_t i);
int32_t     ltr_region_stop(const ltr_region_set* rs, int64_t i);
int32_t     ltr_region_period(const ltr_region_set* rs, int64_t i);
/* FastaReader (src/fasta_reader.h:25-118, .cpp:10-95): one FASTA file with its .fai, or a directory whose *.fa files
 * each have one.  ltr_fasta_fetch = get_sequence(chrom, start, end): 0-based, end inclusive, clamped to the sequence
 * like faidx_fetch_seq; returns the number of bases.  ltr_fasta_seq_len: -1 for an unknown name.
 * ltr_fasta_contig_lines = write_all_contigs_to_vcf (the ##contig header lines). */
typedef struct ltr_fasta ltr_fasta;
int         ltr_fasta_open(const char* path, ltr_fasta** out, char* err, int err_cap);
void        ltr_fasta_close(ltr_fasta* fa);
int64_t     ltr_fasta_num_seqs(const ltr_fasta* fa);
const char* ltr_fasta_seq_name(const ltr_fasta* fa, int64_t i);
int64_t     ltr_fasta_seq_len(const ltr_fasta* fa, const char* chrom);
int64_t     ltr_fasta_fetch(ltr_fasta* fa, const char* chrom, int64_t start, int64_t end, char* out, int64_t cap, char* err, int err_cap);
int64_t     ltr_fasta_contig_lines(const ltr_fasta* fa, char* out, int64_t cap);
/* VCFWriter (src/vcf_writer.h:25-84, .cpp:3-36): records are held in a heap by position and written once no later
 * record of the chromosome can precede them (MAX_RECORD_PAD = 50 bp), all of them at a chromosome change and at
 * close.  A path ending in ".gz" / ".bgz" is written as BGZF like the reference's bgzfostream (src/bgzf_streams.h),
 * any other path as the same text uncompressed.  ltr_vcf_writer_close also frees the writer. */
typedef struct ltr_vcf_writer ltr_vcf_writer;
int ltr_vcf_writer_open(const char* path, ltr_vcf_writer** out);
int ltr_vcf_writer_header(ltr_vcf_writer* w, const char* text);
int ltr_vcf_writer_add_record(ltr_vcf_writer* w, const char* chrom, int32_t record_pos, const char* record_text);
int ltr_vcf_writer_close(ltr_vcf_writer* w);

/* Indexed BAM input without htslib (ltr_bam.cpp): one or more position-sorted *.bam files, each with its *.bam.bai
 * (or *.bai), read as ONE stream like BamCramMultiReader (src/bam_io.h:520-583, src/bam_io.cpp:201-244):
 * merge_by_position != 0 orders the records of all files by position (ORDER_ALNS_BY_POSITION), 0 file by file.
 * ltr_bam_set_region(chrom, start, end) = SetRegion (bam_io.cpp:142-169, 201-220): the records overlapping the 0-based
 * half-open [start, end) (htslib region "chrom:start+1-end"), up to the first one that starts after end + 1.
 * ltr_bam_next = GetNextAlignment (bam_io.cpp:172-197, 222-244) + ExtractSequenceFields (:14-40): 1 = a record, 0 =
 * done; the pointers of the record stay valid until the next ltr_bam_next / ltr_bam_set_region on the handle.
 * The CIGAR comes as the two arrays ltr_alignment wants; quals are Phred + 33.  CRAM is not read. */
typedef struct ltr_bam ltr_bam;
typedef struct ltr_bam_record {
  const char*    name;
  int32_t        file_index;            /* which of the opened files */
  int32_t        ref_id, pos, end_pos;  /* 0-based start; end_pos = bam_endpos (one past the last reference base) */
  int32_t        mapq, flag;
  int32_t        mate_ref_id, mate_pos, tlen;
  int32_t        length;                /* l_seq */
  const char*    bases;                 /* [length] "=ACMGRSVTWYHKDBN" decoding, NUL-terminated */
  const char*    quals;                 /* [length] Phred + 33, NUL-terminated */
  int32_t        n_cigar;
  const char*    cigar_type;            /* [n_cigar] MIDNSHP=X */
  const int32_t* cigar_num;
  const uint8_t* aux; int32_t aux_len;  /* the raw tag block (ltr_bam_aux_*) */
} ltr_bam_record;
int         ltr_bam_open(const char* const* paths, int32_t n_files, int32_t merge_by_position, ltr_bam** out, char* err, int err_cap);
void        ltr_bam_close(ltr_bam* b);
int32_t     ltr_bam_num_refs(const ltr_bam* b);
const char* ltr_bam_ref_name(const ltr_bam* b, int32_t i);
int64_t     ltr_bam_ref_len(const ltr_bam* b, int32_t i);
int32_t     ltr_bam_num_read_groups(const ltr_bam* b);           /* @RG lines of all files (BamHeader::parse_read_groups, bam_io.cpp:43-70) */
const char* ltr_bam_read_group_id(const ltr_bam* b, int32_t i);
const char* ltr_bam_read_group_sample(const ltr_bam* b, int32_t i);
const char* ltr_bam_read_group_library(const ltr_bam* b, int32_t i);
int32_t     ltr_bam_read_group_file(const ltr_bam* b, int32_t i);
int         ltr_bam_set_region(ltr_bam* b, const char* chrom, int32_t start, int32_t end);
int         ltr_bam_next(ltr_bam* b, ltr_bam_record* rec);
/* BamAlignment::GetIntTag / GetFloatTag / GetCharTag / GetStringTag (src/bam_io.h:182-212): 1 = present with that type */
int         ltr_bam_aux_int(const ltr_bam_record* rec, const char tag[2], int64_t* value);
int         ltr_bam_aux_float(const ltr_bam_record* rec, const char tag[2], double* value);
int         ltr_bam_aux_char(const ltr_bam_record* rec, const char tag[2], char* value);
const char* ltr_bam_aux_string(const ltr_bam_record* rec, const char tag[2]);      /* NULL when absent */

/* ---- timers ------------------------------------------------------------------ */
/*
 * The reference's per-genotyper clocks, accumulated per context (wall-clock seconds here, clock() CPU
 * seconds there; reset != 0 zeroes them after the read):
 *   hap_build_s   total_hap_build_time_ (seq_stutter_genotyper.cpp:417,479-480): candidate generation and
 *                 haplotype construction -- ltr_build_haplotype, ltr_haplotype_align_to_ref
 *   hap_aln_s     total_hap_aln_time_ (:515,:561-562): everything inside ltr_process_reads, ltr_align_batch
 *                 and ltr_calc_hap_aln_probs (host preparation, upload, kernels, download, scatter)
 *   posterior_s   total_posterior_time_ (genotyper.cpp:46,:80-81): ltr_posteriors, ltr_plan_posteriors
 *   dp_kernel_ms  device time of the DP kernels inside hap_aln_s (HIP events around every execute)
 *   nw_kernel_ms  device time of the Needleman-Wunsch kernels inside hap_build_s (HIP events, first launch to last)
 *   short_kernel_ms  device time of the seeded stutter path's kernels inside hap_aln_s (same)
 *   dp_cells, dp_pairs  what those DP kernels scored: nominal cells (ltr_plan_cells) and pairs of every plan executed and
 *                 fetched -- for ltr_calc_hap_aln_probs the pairs left AFTER pooling and the de-duplication of trimmed reads,
 *                 i.e. the work the call really did, not reads x haplotypes
 */
typedef struct ltr_timers {
  double  hap_build_s, hap_aln_s, posterior_s, dp_kernel_ms;
  int64_t hap_build_calls, hap_aln_calls, posterior_calls;
  double  nw_kernel_ms, short_kernel_ms;
  double  dp_cells;
  int64_t dp_pairs;
} ltr_timers;
int ltr_ctx_timers(ltr_ctx* ctx, ltr_timers* out, int reset);
/* The same for a caller built against another version of this header: at most out_bytes are written (a longer struct of a
 * newer caller keeps its tail; a shorter one of an older caller is not overrun). */
int ltr_ctx_timers_n(ltr_ctx* ctx, void* out, size_t out_bytes, int reset);
/* Measurement aid: with ltr_ctx_set_debug(ctx, "short_split", 1) the seeded stutter path (HapAligner.cpp:27-233) records events
 * between its launches; out_ms = device milliseconds summed over the calls since the last reset: [0] quality tables + flank rows
 * before the stutter block (:36-44, :112-159), [1] the stutter-block row (:64-111, StutterAlignerClass.cpp:59-154), [2] flank rows
 * after it, [3] the seed log-sum (compute_aln_logprob, :165-233). */
int ltr_ctx_short_kernel_split(ltr_ctx* ctx, double out_ms[4], int reset);

/* ---- host threads -------------------------------------------------------------- */
/*
 * The reference is one thread per process and N processes per node (README.md:78-82).  This library's host loops -- pooling,
 * trimming and planning inside ltr_calc_hap_aln_probs (seq_stutter_genotyper.cpp:514-563), the string passes of the NW call --
 * run on worker threads of the PROCESS (two pools), never more than the budget each:
 *   rule = min(CPUs of the affinity mask, cgroup CPU quota, hardware threads) / ranks on this host, clamped to 1 .. 16,
 * ranks on this host = LOCAL_WORLD_SIZE of the environment (torch.distributed.run sets it) or 1.  The helper thread that stages
 * the next chunk of ltr_calc_hap_aln_probs while the current one is planned is used from a budget of 12 up (below that two
 * thread teams on a handful of cores lose to one).  Results never depend on the budget.
 *   ltr_ctx_set_host_threads   n >= 1: the budget (per process: every context shares the pools); 0: back to the rule
 *   ltr_ctx_host_threads       the budget in force
 *   ltr_host_threads_rule      what the rule gives for `local_world_size` ranks on this host (<= 0: LOCAL_WORLD_SIZE); needs no
 *                              context: a launcher that knows its rank count calls ltr_ctx_set_host_threads(ctx, ltr_host_threads_rule(n))
 */
int ltr_ctx_set_host_threads(ltr_ctx* ctx, int n);
int ltr_ctx_host_threads(const ltr_ctx* ctx);
int ltr_host_threads_rule(int local_world_size);

/* ABI of this header: bumped whenever a struct grows or a capacity contract changes (6: ltr_timers carries dp_cells / dp_pairs
 * since round 5; ltr_plan_kernel_ranges may return up to ltr_num_kernels() ranges; round 6 added the functions above and
 * ltr_ctx_wg_first_pass).  A caller checks ltr_abi_version() == LTR_ABI_VERSION before passing structs. */
#define LTR_ABI_VERSION 6
int ltr_abi_version(void);
const char* ltr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LTR_GPU_H_ */
