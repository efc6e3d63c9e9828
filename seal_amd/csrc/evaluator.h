// Host orchestration: the device-resident counterparts of seal::Ciphertext, seal::KSwitchKeys and
// seal::Evaluator for the hot path.  Same method names, argument meaning, metadata updates and
// exception classes as the reference (native/src/seal/evaluator.h:79-1387, evaluator.cpp); the
// arithmetic is enqueued on a HIP stream as the kernels of ntt_kernels.hip / poly_kernels.hip /
// behz_kernels.hip.
#pragma once
#include "context.h"
#include "poly_kernels.h"
#include "pool.h"
#include "comm.h"
#include "ntt2_kernels.h"
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <vector>

namespace sealhip
{
    // A batch of `batch` ciphertexts sharing metadata; slab layout [poly][batch][K][N].
    class Evaluator;
    // A key-switch tail that has not run yet (Evaluator::switch_key_inplace and the digit-parallel switch_key_finish; CKKS and BFV at
    // the two-pass sizes): the two planes of the ciphertext still hold the addends, `acc` the key-switch sums [batch][2][K+1][N].
    // Whoever touches the words next completes it - except, on the same evaluator, CKKS rescale_to_next_inplace and BFV
    // mod_switch_to_next_inplace, which fold the mod-down and their own division into one pass (ntt_kernels.h: NttTail2;
    // switch_key_finish_modswitch_bfv).  Results are the same words either way.
    struct LazyTail
    {
        const Evaluator *owner;
        uint64_t *acc;
        // CKKS on the fused path: the data-prime components of acc are c + S P^-1 already (ntt2_kernels.h: KsFusedArgs::fold_c0) -
        // the tail, folded into a rescale or not, reads one operand where it read two
        bool with_addend = false;
    };
    // A CKKS 2 x 2 tensor product that has not been formed yet (round 6; Evaluator::multiply into a third object, multiply_inplace and
    // square_inplace; batches large enough for the un-split key switch, two-pass sizes): the destination has its shape and metadata, its
    // words are pending.  A
    // relinearize_inplace on the same evaluator then never stores the product: the third polynomial is formed inside the inverse
    // transform that opens the key switch (NttBatch::prod_x), the first two inside the key switch's last epilogue
    // (KsFusedArgs::fold_x) - one kernel and 45 MB per ciphertext less at the headline size.  Anything else that touches the
    // destination's words forms the product first (Ciphertext::settle), and anything that is about to WRITE, re-shape or destroy an
    // operand forms the products that read it first (Ciphertext::before_write): same words either way.  SEALHIP_LAZY_PRODUCT=0 turns
    // the deferral off.
    class Ciphertext;
    struct LazyProduct
    {
        const Evaluator *owner;
        // operands that are live objects; null = that operand's two polynomials are in `own`
        const Ciphertext *x, *y;
        // in-place products (multiply_inplace, square_inplace): the destination's PREVIOUS slab - its two polynomials as they were -
        // belongs to the record until the product is formed, consumed or discarded; the destination itself got a fresh slab of three
        uint64_t *own;
        const uint64_t *xw() const; // plane 0 of operand x (settles whatever is pending on a live operand)
        const uint64_t *yw() const;
    };
    void lazy_product_stats(uint64_t &fused, uint64_t &formed, uint64_t &dropped);
    // rotations that read their operand through the automorphism's index map inside the key switch / that ran the permutation kernels
    void galois_path_stats(uint64_t &gathered, uint64_t &permuted);
    // process-wide counters (tests, tools): tails folded into a rescale / completed on their own / discarded unrun
    void lazy_tail_stats(uint64_t &folded, uint64_t &plain, uint64_t &dropped);
    // chunked key switching (evaluator_keyswitch.cpp): calls that ran in chunks, chunks issued, largest intermediate held since the previous query (words)
    void ks_chunk_stats(uint64_t *calls, uint64_t *chunks, uint64_t *scratch_words_max);

    class Ciphertext
    {
    public:
        explicit Ciphertext(const Context &ctx, size_t batch = 1) : ctx_(&ctx), batch_(batch ? batch : 1) {}
        ~Ciphertext();
        Ciphertext(const Ciphertext &o);
        Ciphertext &operator=(const Ciphertext &o);

        const Context &context() const { return *ctx_; }
        size_t batch() const { return batch_; }
        size_t size() const { return size_; }
        const Level *level() const { return level_; }
        size_t coeff_modulus_size() const { return level_ ? level_->K : 0; }
        size_t poly_modulus_degree() const { return level_ ? ctx_->n() : 0; }
        bool &is_ntt_form() { return is_ntt_form_; }
        bool is_ntt_form() const { return is_ntt_form_; }
        double &scale() { return scale_; }
        double scale() const { return scale_; }
        uint64_t &correction_factor() { return correction_factor_; }
        uint64_t correction_factor() const { return correction_factor_; }
        // every access to the words completes a deferred key-switch tail first (LazyTail)
        uint64_t *data()
        {
            settle();
            before_write(); // a mutable pointer is a write as far as pending products of THIS ciphertext's words are concerned
            return data_;
        }
        const uint64_t *data() const
        {
            settle();
            return data_;
        }
        size_t plane_words() const { return level_ ? batch_ * level_->K * ctx_->n() : 0; }
        size_t word_count() const { return size_ * plane_words(); }
        uint64_t *plane(size_t p) { return data() + p * plane_words(); }
        const uint64_t *plane(size_t p) const { return data() + p * plane_words(); }
        bool has_lazy_tail() const { return lazy_ != nullptr; }
        bool has_lazy_product() const { return lazy_prod_ != nullptr; }
        bool has_storage() const { return data_ != nullptr; } // a question about the buffer, not about its words

        // Ciphertext::resize(context, parms_id, size) (ciphertext.cpp:101-116): keeps the leading
        // polynomials when only `size` changes at the same level; contents are unspecified after a
        // level change.  New polynomials are zero (the reference's DynArray zero-fills).
        void resize(const Level *level, size_t size, hipStream_t stream);
        // adopt a freshly computed slab (takes ownership of `words` from the pool)
        void adopt(const Level *level, size_t size, uint64_t *slab, size_t capacity_words);
        // same shape change as resize() but the contents are left undefined (the caller overwrites them)
        void reshape_uninitialized(const Level *level, size_t size);
        size_t capacity_words() const { return capacity_words_; }
        void release();
        // Ciphertext::reserve (ciphertext.cpp:38-78): room for `size_capacity` polynomials at `level`; the size shrinks to the capacity when
        // it was larger, the leading words are kept (at the same level they are the leading polynomials).  2 <= size_capacity <= 16.
        void reserve(const Level *level, size_t size_capacity, hipStream_t stream);
        // Ciphertext::size_capacity (ciphertext.h:520-528): polynomials the slab has room for at the current level
        size_t size_capacity() const { return plane_words() ? capacity_words_ / plane_words() : 0; }
        // sealc Ciphertext_SetParmsId writes the member without any check (c/ciphertext.cpp:239-247); here a level is a pointer into the context
        void set_level_unchecked(const Level *level)
        {
            settle();
            before_write();
            level_ = level;
        }

    private:
        const Context *ctx_;
        size_t batch_;
        size_t size_ = 0;
        const Level *level_ = nullptr;
        bool is_ntt_form_ = false;
        double scale_ = 1.0;
        uint64_t correction_factor_ = 1;
        uint64_t *data_ = nullptr;
        size_t capacity_words_ = 0;
        // deferred key-switch tail (owned; see LazyTail)
        mutable LazyTail *lazy_ = nullptr;
        void settle() const; // run it now
        void drop_lazy();    // the words are about to be replaced: discard it
        // pending tensor product whose destination this is (owned; see LazyProduct), and the pending products that READ this object
        mutable LazyProduct *lazy_prod_ = nullptr;
        mutable std::vector<const Ciphertext *> prod_readers_;
        mutable unsigned prod_reader_count_ = 0; // = prod_readers_.size(), readable without the list's lock
        void settle_product() const;             // form it now
        void drop_product();                     // the words are about to be replaced: discard it
        void before_write() const                // form every pending product that reads this object's words
        {
            if (__atomic_load_n(&prod_reader_count_, __ATOMIC_ACQUIRE))
                settle_readers();
        }
        void settle_readers() const;
        // the slab is replaced by `slab` (uninitialised, room for capacity_words) and handed to the caller instead of the pool
        uint64_t *exchange_slab(const Level *level, size_t size, uint64_t *slab, size_t capacity_words);
        friend class Evaluator;
        friend struct LazyProduct;
        friend void lazy_product_link(const Ciphertext *, const LazyProduct &);
        friend void lazy_product_unlink(const Ciphertext *, const LazyProduct &);
    };

    // LazyProduct bookkeeping (objects.cpp): the reader lists of the live operands, only touched under one global lock
    void lazy_product_link(const Ciphertext *dest, const LazyProduct &p);
    void lazy_product_unlink(const Ciphertext *dest, const LazyProduct &p);

    // seal::Plaintext (plaintext.h) resident in HBM: either coeff_count <= N coefficients modulo t (BFV/BGV,
    // parms_id_zero) or, in NTT form, K*N words at a level (CKKS always; BFV/BGV after transform_to_ntt_inplace).
    // One plaintext is applied to every item of a ciphertext batch.
    class Plaintext
    {
    public:
        explicit Plaintext(const Context &ctx) : ctx_(&ctx) {}
        ~Plaintext();
        Plaintext(const Plaintext &o);
        Plaintext &operator=(const Plaintext &o);

        const Context &context() const { return *ctx_; }
        size_t coeff_count() const { return coeff_count_; }
        const Level *level() const { return level_; } // parms_id: null = parms_id_zero = not in NTT form
        void set_level(const Level *l) { level_ = l; }
        bool is_ntt_form() const { return level_ != nullptr; }
        double &scale() { return scale_; }
        double scale() const { return scale_; }
        uint64_t *data() { return data_; }
        const uint64_t *data() const { return data_; }
        // Plaintext::resize (plaintext.h:268-290): new coefficients are zero
        void resize(size_t coeff_count, hipStream_t stream);
        // Plaintext_Set4: coefficient form, `count` words from host or device memory
        void set(const uint64_t *words, size_t count, bool from_device);
        void adopt(uint64_t *slab, size_t count, size_t capacity_words);
        void adopt_count(size_t count) { coeff_count_ = count; } // shrink in place (plaintext mod switching)

    private:
        const Context *ctx_;
        size_t coeff_count_ = 0;
        const Level *level_ = nullptr;
        double scale_ = 1.0;
        uint64_t *data_ = nullptr;
        size_t capacity_words_ = 0;
    };

    // KSwitchKeys::keys_ (kswitchkeys.h:340): per index, `digits` size-2 key-level ciphertexts in NTT
    // form, stored as one slab [digit][2][L][N].
    class KSwitchKeys
    {
    public:
        struct Key
        {
            uint64_t *dev = nullptr;
            size_t digits = 0; // decomposition digits resident in `dev`
            size_t digit0 = 0; // index of the first of them (non-zero only for a digit-parallel slice)
            // true: every component is stored in the register order of the fused key-switch
            // kernel, as doubles for primes of the double-precision back end (ntt2_kernels.h)
            bool register_order = false;
        };
        ~KSwitchKeys();
        // words: `digits` digits starting at digit `digit0` of key `index`, each 2 x L x N words
        void set_key(const Context &ctx, size_t index, size_t digits, const uint64_t *words, bool from_device, size_t digit0 = 0);
        // the same with the words delivered by `upload(device_destination)` (e.g. piecewise from a serialized stream)
        void set_key_with(const Context &ctx, size_t index, size_t digits, const std::function<void(uint64_t *)> &upload, size_t digit0 = 0);
        // the words of key `index` as set_key took them ([digits][2][L][N], canonical, natural order), written to device memory
        void key_words(size_t index, uint64_t *device_out) const;
        // one digit of key `index` ([2][L][N] words, as key_words): the transient conversion buffer of a save is one digit, not a key
        void digit_words(size_t index, size_t digit, uint64_t *device_out) const;
        // HBM held by the keys (register order: ntt2_kernels.h - a C5 key is 284 MB, its natural words 252 MB)
        size_t device_bytes() const;
        void clear(); // drop every key (KSwitchKeys::load replaces the whole object, kswitchkeys.cpp:92-180)
        bool has_key(size_t index) const { return index < keys_.size() && keys_[index].dev != nullptr; }
        const Key &key(size_t index) const { return keys_[index]; }
        size_t slots() const { return keys_.size(); }
        // KSwitchKeys::data().size() of the reference object: N for GaloisKeys (galoiskeys.h), what a stream said for a loaded one
        void reserve_slots(size_t count)
        {
            if (keys_.size() < count)
                keys_.resize(count);
        }
        size_t size() const;
        const Context *context() const { return ctx_; }
        // KSwitchKeys::operator= (kswitchkeys.h:74-101): a deep copy, every key slab duplicated in HBM
        void assign(const KSwitchKeys &other);
        // KSwitchKeys::parms_id (kswitchkeys.h:169-183): the key level's once a key is installed, unless the caller wrote another
        void get_parms_id(uint64_t *out) const;
        void set_parms_id(const uint64_t *id);

    private:
        std::vector<Key> keys_;
        const Context *ctx_ = nullptr;
        uint64_t parms_id_[4] = { 0, 0, 0, 0 };
        bool parms_id_written_ = false;
        size_t key_bytes(const Key &k) const;
    };

    class Evaluator
    {
    public:
        explicit Evaluator(const Context &context);
        ~Evaluator();

        const Context &context() const { return context_; }
        // work of this evaluator is enqueued on `s` (nullptr = the NULL stream); the stream is registered with the device
        // pool so that memory recycled between evaluators on different streams is ordered (pool.h)
        void set_stream(hipStream_t s);
        hipStream_t stream() const { return stream_; }
        void synchronize() const;
        // SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT (evaluator.cpp:386-392) needs a device->host round trip
        // per operation; off by default for device-resident batches, on for drop-in parity checks.
        void set_transparent_check(bool on) { transparent_check_ = on; }

        void negate_inplace(Ciphertext &encrypted) const;
        void add_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const;
        void sub_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const;
        void multiply_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const;
        void multiply(const Ciphertext &encrypted1, const Ciphertext &encrypted2, Ciphertext &destination) const;
        void square_inplace(Ciphertext &encrypted) const;
        void relinearize_inplace(Ciphertext &encrypted, const KSwitchKeys &relin_keys) const;
        void mod_switch_to_next_inplace(Ciphertext &encrypted) const;
        void mod_switch_to_inplace(Ciphertext &encrypted, const uint64_t *parms_id) const;
        void rescale_to_next_inplace(Ciphertext &encrypted) const;
        void rescale_to_inplace(Ciphertext &encrypted, const uint64_t *parms_id) const;
        void mod_reduce_to_next_inplace(Ciphertext &encrypted) const;
        void mod_reduce_to_inplace(Ciphertext &encrypted, const uint64_t *parms_id) const;
        void transform_to_ntt_inplace(Ciphertext &encrypted) const;
        // plaintext operands and the many-operand forms (evaluator.cpp:242-261, 1649-2287, 1369-1402)
        void add_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const;
        void sub_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const;
        void multiply_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const;
        void transform_to_ntt_inplace(Plaintext &plain, const uint64_t *parms_id) const;
        void mod_switch_to_next_inplace(Plaintext &plain) const;
        void mod_switch_to_inplace(Plaintext &plain, const uint64_t *parms_id) const;
        void add_many(const std::vector<const Ciphertext *> &encrypteds, Ciphertext &destination) const;
        void multiply_many(const std::vector<const Ciphertext *> &encrypteds, const KSwitchKeys &relin_keys, Ciphertext &destination) const;
        void exponentiate_inplace(Ciphertext &encrypted, uint64_t exponent, const KSwitchKeys &relin_keys) const;
        void transform_from_ntt_inplace(Ciphertext &encrypted_ntt) const;
        void apply_galois_inplace(Ciphertext &encrypted, uint32_t galois_elt, const KSwitchKeys &galois_keys) const;
        // out-of-place forms (evaluator.h:1072-1315 of the reference): with the exact Galois key present `encrypted` is read where
        // it lies - no copy into the destination first
        void apply_galois(const Ciphertext &encrypted, uint32_t galois_elt, const KSwitchKeys &galois_keys, Ciphertext &destination) const;
        void rotate_rows(const Ciphertext &encrypted, int steps, const KSwitchKeys &galois_keys, Ciphertext &destination) const;
        void rotate_columns(const Ciphertext &encrypted, const KSwitchKeys &galois_keys, Ciphertext &destination) const;
        void rotate_vector(const Ciphertext &encrypted, int steps, const KSwitchKeys &galois_keys, Ciphertext &destination) const;
        void complex_conjugate(const Ciphertext &encrypted, const KSwitchKeys &galois_keys, Ciphertext &destination) const;
        void rotate_rows_inplace(Ciphertext &encrypted, int steps, const KSwitchKeys &galois_keys) const;
        void rotate_columns_inplace(Ciphertext &encrypted, const KSwitchKeys &galois_keys) const;
        void rotate_vector_inplace(Ciphertext &encrypted, int steps, const KSwitchKeys &galois_keys) const;
        void complex_conjugate_inplace(Ciphertext &encrypted, const KSwitchKeys &galois_keys) const;

        bool is_transparent(const Ciphertext &ct) const; // synchronises

        // hipGraph capture of a fixed operation sequence.  Between begin_capture() and end_capture() the
        // operations issued on this evaluator are RECORDED (stream capture, side streams included), not executed; the returned
        // executable graph replays them with one launch on the same device buffers - operands, destinations and scratch keep
        // the addresses they had during capture, so the caller refreshes the operands' contents in place and reads the
        // destinations after each replay.  Run the sequence once eagerly first (lazily built tables, pool warm-up), keep the
        // captured objects alive and do not resize them between replays.  At small batches the step is launch-bound on the
        // host (about 25 kernel launches + stream fork/join per multiply+relinearize+rescale): see profiles/HISTORY.md section 5 ("Small batches / latency").
        // an executable graph together with the scratch blocks whose addresses it replays on (kept out of the pool until
        // the graph is destroyed)
        struct Graph
        {
            hipGraphExec_t exec = nullptr;
            std::vector<uint64_t *> scratch;
            ~Graph();
        };
        void begin_capture();
        Graph *end_capture();
        void launch_graph(const Graph *graph) const;

        static size_t relin_index(size_t key_power);       // RelinKeys::get_index  (relinkeys.h:58)
        static size_t galois_index(uint32_t galois_elt);   // GaloisKeys::get_index (galoiskeys.h:48)
        uint32_t galois_elt_from_step(int step) const;      // GaloisTool::get_elt_from_step (galois.cpp:53)

        // switch_key_inplace (evaluator.cpp:2561): encrypted (size >= 2) += KS(target), target = one
        // plane [batch][K][N] in the scheme's native form.
        // c1_zero_unwritten: the caller's ciphertext is (c0, 0) and its second polynomial has NOT been written (rotations); it is zeroed here
        // only when somebody is going to read it
        // galois_elt / galois_c0 (round 6, rotations at the batches that fold the addend into un-split sums, ks_gathers()): `target` and
        // galois_c0 are the UNPERMUTED c1 and c0 of the operand, read through the NTT-domain automorphism inside the key switch's
        // kernels; encrypted's own polynomials are then never read
        void switch_key_inplace(Ciphertext &encrypted, const uint64_t *target, const KSwitchKeys &keys, size_t key_index, bool c1_zero_unwritten = false,
                                uint32_t galois_elt = 0, const uint64_t *galois_c0 = nullptr) const;
        // the rules of switch_key_inplace, for callers that prepare their operands differently per path
        unsigned ks_split(const Ciphertext &encrypted, const KSwitchKeys &keys, size_t key_index) const;
        bool ks_folds(const KSwitchKeys &keys, size_t key_index, unsigned K) const;
        // the two halves of switch_key_inplace for digit-parallel key switching over several GPUs:
        // acc = [batch][2][K+1][N] words (switch_key_acc_words); partial fills it with the canonical partial sums of
        // the digits [j0, j1); finish reduces the sum of `parts` such buffers and applies the mod-down to encrypted.
        size_t switch_key_acc_words(const Ciphertext &encrypted) const;
        // split > 1 (fused path only): the digit range is cut into `split` in-launch groups and acc holds `split` buffers
        // (ntt2_kernels.h: KsFusedArgs::parts); the caller adds them with k_keyswitch_reduce(..., local_parts = split)
        // product (round 6, CKKS, fused path, split 1, fold_addend): `encrypted` is a tensor product that was never stored - `target`
        // is where its third polynomial is WRITTEN (by the inverse transform that forms it from the operands), the two leading
        // polynomials are formed in the key switch's epilogue (KsFusedArgs::fold_x)
        void switch_key_partial(const Ciphertext &encrypted, const uint64_t *target, const KSwitchKeys &keys, size_t key_index,
                                unsigned j0, unsigned j1, uint64_t *acc, unsigned split = 1, bool fold_addend = false,
                                const LazyProduct *product = nullptr, bool addend1_zero = false, uint32_t galois_elt = 0,
                                const uint64_t *galois_c0 = nullptr) const;
        // fold_addend (CKKS, fused path, the full digit range, split 1): the data-prime components of acc leave as c + S P^-1
        // (KsFusedArgs::fold_c0); the matching finish call says so with acc_has_addend
        // may_defer (relinearize_finish / apply_galois_finish / the in-library exchange): CKKS at the two-pass sizes copies the reduced
        // sums - with the ciphertext's words added - into a block of its own and leaves the mod-down pending like switch_key_inplace
        // does (LazyTail), so that a rescale that follows folds both divisions; `acc` is not referenced after the call returns
        void switch_key_finish(Ciphertext &encrypted, uint64_t *acc, unsigned parts, bool acc_has_addend = false, bool may_defer = false) const;
        // relinearize (size 3 -> 2) and apply_galois (size 2) split the same way: *_partial leaves `encrypted` ready for
        // the finish call (for apply_galois: c0 <- pi(c0), c1 <- 0) and writes this rank's partial sums to acc
        void relinearize_partial(Ciphertext &encrypted, const KSwitchKeys &relin_keys, unsigned j0, unsigned j1, uint64_t *acc) const;
        void relinearize_finish(Ciphertext &encrypted, uint64_t *acc, unsigned parts) const;
        void apply_galois_partial(Ciphertext &encrypted, uint32_t galois_elt, const KSwitchKeys &galois_keys, unsigned j0, unsigned j1,
                                  uint64_t *acc) const;
        void apply_galois_finish(Ciphertext &encrypted, uint64_t *acc, unsigned parts) const;

        // ---- digit-parallel key switching with the exchange INSIDE the library: RCCL calls on this
        // evaluator's stream, no host synchronisation.  Every rank of `comm` calls the same method with equal ciphertexts and
        // a key that holds (at least) its own digits comm_split(K, size, rank).  Two exchange shapes:
        //   all_reduce      one all-reduce of the 2(K+1)N partial sums per ciphertext; every rank runs the whole mod-down
        //   reduce_scatter  (CKKS) the sums of the K data moduli are reduce-scattered by owner (comm_split(K, size, rank)),
        //                   the special-prime component is all-reduced, every rank runs the mod-down of ITS moduli only and
        //                   the increments are all-gathered: same bytes on the wire, 1/size of the mod-down per rank
        // Results are bit-identical to the single-GPU operations.  BFV / BGV use the all-reduce shape in either mode.
        enum class KsExchange
        {
            all_reduce = 0,
            reduce_scatter = 1,
        };
        void relinearize_inplace(Ciphertext &encrypted, const KSwitchKeys &relin_keys, Comm &comm, KsExchange how) const;
        void apply_galois_inplace(Ciphertext &encrypted, uint32_t galois_elt, const KSwitchKeys &galois_keys, Comm &comm, KsExchange how) const;
        void rotate_vector_inplace(Ciphertext &encrypted, int steps, const KSwitchKeys &galois_keys, Comm &comm, KsExchange how) const;
        // one-time distribution of a key-switching key: rank `root` holds the full key `index`, every rank ends up with its own
        // digits resident (the key is broadcast in its natural layout and re-laid out locally)
        // (staging: a device buffer of K * 2 * L * N words on EVERY rank; on `root` it holds the key [digit][2][L][N])
        void broadcast_key_digits(KSwitchKeys &keys, size_t index, uint64_t *staging, Comm &comm, int root) const;
        // the three local phases of the reduce-scatter shape (the parity tests emulate the ranks in one process):
        //   slots          m = ceil(K / nranks) moduli slots per rank
        //   pack_targets   acc (this rank's partial sums) -> send [nranks][m][batch][2][N] + sp [batch][2][N] (special prime)
        //   finish_owned   recv = the summed chunk of `rank` [m][batch][2][N], sp = the summed special component
        //                  -> own [m][batch][2][N]: the increments of this rank's moduli
        //   add_gathered   all [nranks][m][batch][2][N] -> encrypted += increments
        unsigned switch_key_slots(const Ciphertext &encrypted, unsigned nranks) const;
        void switch_key_pack_targets(const Ciphertext &encrypted, const uint64_t *acc, unsigned nranks, uint64_t *send, uint64_t *sp) const;
        void switch_key_finish_owned(
            const Ciphertext &encrypted, const uint64_t *recv, const uint64_t *sp, unsigned nranks, unsigned rank, uint64_t *own) const;
        void switch_key_add_gathered(Ciphertext &encrypted, const uint64_t *all, unsigned nranks) const;

    private:
        friend class Encryptor; // public-key encryption ends with one modulus switch from the level above (encryptor.cpp:139-186)
        void check_valid(const Ciphertext &ct, const char *what) const;
        bool scale_within_bounds(double scale, const Level &lvl) const;
        void bfv_multiply(Ciphertext &e1, const Ciphertext &e2) const;
        void bfv_multiply_to(const Ciphertext &e1, const Ciphertext &e2, Ciphertext &dst) const;
        void bgv_multiply(Ciphertext &e1, const Ciphertext &e2) const;
        void check_valid(const Plaintext &plain) const;
        void addsub_plain(Ciphertext &encrypted, const Plaintext &plain, int op) const;
        bool mul_plain_monomial(Ciphertext &encrypted, const Plaintext &plain) const;
        void plain_to_rns(const Plaintext &plain, const Level &lvl, uint64_t scale_by, uint64_t *out) const;
        void multiply_plain_ntt(Ciphertext &encrypted_ntt, const uint64_t *plain_rns, const Level *plain_level, double plain_scale) const;
        void bgv_correct_and_combine(
            Scratch &delta, const uint64_t *a, size_t a_stride, const ShoupOp *mul, unsigned ncomp, size_t items, uint64_t *out0,
            uint64_t *out1, size_t out_stride, int epi) const;
        void ckks_multiply(Ciphertext &e1, const Ciphertext &e2) const;
        void tensor_2x2(Ciphertext &e1, const Ciphertext &e2, const Level &lvl, const struct PlaneGeom &g) const;
        void mod_switch_scale_to_next(Ciphertext &encrypted) const;
        void mod_switch_drop_to_next(Ciphertext &encrypted) const;
        void rotate_internal(Ciphertext &encrypted, int steps, const KSwitchKeys &galois_keys) const;
        void rotate_internal(const Ciphertext &encrypted, int steps, const KSwitchKeys &galois_keys, Ciphertext &destination) const;
        void conjugate_internal(Ciphertext &encrypted, const KSwitchKeys &galois_keys) const;
        void throw_if_transparent(const Ciphertext &ct) const;
        void switch_key_exchange_finish(Ciphertext &encrypted, uint64_t *acc, Comm &comm, KsExchange how) const;
        const uint32_t *ks_comp_prime(unsigned K) const;
        int ks_class_hint(unsigned K) const;
        struct KsTargets
        {
            uint32_t *dev = nullptr; // [targets1: int, fp | targets2: int, fp]
            unsigned n_int = 0, n_fp = 0;
        };
        const KsTargets &ks_targets(unsigned K) const;

        const Context &context_;
        hipStream_t stream_ = nullptr;
        hipStream_t capture_stream_ = nullptr, saved_stream_ = nullptr;
        bool capturing_ = false;
        bool transparent_check_ = false;
        // lazily built per-level tables; guarded so that concurrent calls on different ciphertexts stay safe, as with the
        // reference's Evaluator (evaluator.h:79-87: the class holds only immutable state)
        // ciphertexts whose key-switch tail this evaluator deferred (LazyTail): completed before the evaluator goes away or starts
        // recording a graph
        friend class Ciphertext;
        void defer_tail(Ciphertext &e, uint64_t *acc, bool with_addend) const;
        void complete_tail(Ciphertext &e, LazyTail t) const;     // the plain mod-down, then the sums go back to the pool
        // deferred tensor products (LazyProduct): record / form now / take over for the fused relinearisation / discard
        bool may_defer_product(const Level &lvl, size_t batch) const;
        void defer_product(Ciphertext &dest, const Ciphertext *x, const Ciphertext *y, uint64_t *own) const;
        void complete_product(Ciphertext &dest, LazyProduct p) const;
        LazyProduct detach_product(Ciphertext &dest) const;
        void forget_product(const Ciphertext &dest, LazyProduct p) const;
        bool relinearize_from_product(Ciphertext &e, const KSwitchKeys &relin_keys) const; // false: conditions not met, nothing done
        void forget_tail(const Ciphertext &e, LazyTail t) const; // discard
        LazyTail detach_tail(Ciphertext &e) const;
        void settle_all() const;
        // mod-down by the special prime and rescale by q_last in one pass (NttTail2); e has two polynomials and a deferred tail
        void switch_key_finish_rescale(Ciphertext &e, uint64_t *acc, const Level *next, double destination_scale, bool acc_has_addend) const;
        void switch_key_finish_modswitch_bfv(Ciphertext &e, uint64_t *acc, const Level *next) const;
        mutable std::mutex lazy_mu_;
        mutable std::vector<const Ciphertext *> lazy_cts_;
        mutable std::mutex cache_mu_;
        mutable std::map<unsigned, uint32_t *> ks_maps_;
        mutable std::map<unsigned, KsTargets> ks_targets_;
        mutable unsigned *d_flag_ = nullptr;
    };
} // namespace sealhip
