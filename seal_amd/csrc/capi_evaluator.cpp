// extern "C" layer, part 3: KSwitchKeys, Evaluator (ciphertext, plaintext-operand and many-operand forms), communicator and digit-parallel forms (include/sealhip.h)
#include "capi_common.h"

extern "C"
{
    // ------------------------------------------------------------------ KSwitchKeys
    SHL_FUNC KSwitchKeys_Create1(void **kswitch_keys)
    {
        IfNullRet(kswitch_keys, SHL_E_POINTER);
        SHL_TRY
        *kswitch_keys = new KSwitchKeys();
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<KSwitchKeys>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC KSwitchKeys_Size(void *thisptr, uint64_t *size)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(size, SHL_E_POINTER);
        *size = as<KSwitchKeys>(thisptr)->size();
        return SHL_S_OK;
    }
    SHL_FUNC KSwitchKeys_SetKey(void *thisptr, void *context, uint64_t index, uint64_t digits, const uint64_t *host_words)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(host_words, SHL_E_POINTER);
        SHL_TRY
        as<KSwitchKeys>(thisptr)->set_key(*as<Context>(context), index, digits, host_words, false);
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_SetKeyFromDevice(void *thisptr, void *context, uint64_t index, uint64_t digits, const uint64_t *device_words)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(device_words, SHL_E_POINTER);
        SHL_TRY
        as<KSwitchKeys>(thisptr)->set_key(*as<Context>(context), index, digits, device_words, true);
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_SetKeyDigits(void *thisptr, void *context, uint64_t index, uint64_t digit_first, uint64_t digits,
                                      const uint64_t *host_words)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(host_words, SHL_E_POINTER);
        SHL_TRY
        as<KSwitchKeys>(thisptr)->set_key(*as<Context>(context), index, digits, host_words, false, digit_first);
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_HasKey(void *thisptr, uint64_t index, bool *has_key)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(has_key, SHL_E_POINTER);
        *has_key = as<KSwitchKeys>(thisptr)->has_key(index);
        return SHL_S_OK;
    }
    SHL_FUNC RelinKeys_GetIndex(uint64_t key_power, uint64_t *index)
    {
        IfNullRet(index, SHL_E_POINTER);
        SHL_TRY
        *index = Evaluator::relin_index(key_power);
        SHL_CATCH
    }
    SHL_FUNC GaloisKeys_GetIndex(uint32_t galois_elt, uint64_t *index)
    {
        IfNullRet(index, SHL_E_POINTER);
        SHL_TRY
        *index = Evaluator::galois_index(galois_elt);
        SHL_CATCH
    }
    SHL_FUNC GaloisTool_GetEltFromStep(void *context, int step, uint32_t *galois_elt)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(galois_elt, SHL_E_POINTER);
        SHL_TRY
        Evaluator ev(*as<Context>(context));
        *galois_elt = ev.galois_elt_from_step(step);
        SHL_CATCH
    }

    // ------------------------------------------------------------------ Evaluator
    SHL_FUNC Evaluator_Create(void *context, void **evaluator)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(evaluator, SHL_E_POINTER);
        SHL_TRY
        *evaluator = new Evaluator(*as<Context>(context));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<Evaluator>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC Evaluator_SetStream(void *thisptr, void *hip_stream)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        as<Evaluator>(thisptr)->set_stream((hipStream_t)hip_stream);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_BeginCapture(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        as<Evaluator>(thisptr)->begin_capture();
        SHL_CATCH
    }
    SHL_FUNC Evaluator_EndCapture(void *thisptr, void **graph)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(graph, SHL_E_POINTER);
        SHL_TRY
        *graph = as<Evaluator>(thisptr)->end_capture();
        SHL_CATCH
    }
    SHL_FUNC Evaluator_LaunchGraph(void *thisptr, void *graph)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(graph, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->launch_graph(static_cast<const Evaluator::Graph *>(graph));
        SHL_CATCH
    }
    SHL_FUNC Graph_Destroy(void *graph)
    {
        IfNullRet(graph, SHL_E_POINTER);
        delete static_cast<Evaluator::Graph *>(graph); // its scratch blocks return to the pool
        return SHL_S_OK;
    }
    SHL_FUNC Evaluator_SetTransparentCheck(void *thisptr, bool enabled)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        as<Evaluator>(thisptr)->set_transparent_check(enabled);
        return SHL_S_OK;
    }
    // destination := encrypted on the evaluator's stream (a pipeline with several evaluators / streams copies its inputs in
    // stream order; Ciphertext_Set works on the calling thread's stream)
    SHL_FUNC Evaluator_CopyTo(void *thisptr, void *encrypted, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        if (encrypted != destination)
            *as<Ciphertext>(destination) = *as<Ciphertext>(encrypted);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Synchronize(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->synchronize();
        SHL_CATCH
    }
#define EV_UNARY(fn, call)                                           \
    SHL_FUNC fn(void *thisptr, void *encrypted, void *destination)   \
    {                                                                \
        IfNullRet(thisptr, SHL_E_POINTER);                           \
        IfNullRet(encrypted, SHL_E_POINTER);                         \
        IfNullRet(destination, SHL_E_POINTER);                       \
        SHL_TRY                                                      \
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());   \
        auto ev = as<Evaluator>(thisptr);                            \
        Ciphertext &d = prepare_dest(encrypted, destination);        \
        ev->call(d);                                                 \
        SHL_CATCH                                                    \
    }
#define EV_UNARY_POOL(fn, call)                                                  \
    SHL_FUNC fn(void *thisptr, void *encrypted, void *destination, void *pool)   \
    {                                                                            \
        (void)pool;                                                              \
        IfNullRet(thisptr, SHL_E_POINTER);                                       \
        IfNullRet(encrypted, SHL_E_POINTER);                                     \
        IfNullRet(destination, SHL_E_POINTER);                                   \
        SHL_TRY                                                                  \
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());               \
        auto ev = as<Evaluator>(thisptr);                                        \
        Ciphertext &d = prepare_dest(encrypted, destination);                    \
        ev->call(d);                                                             \
        SHL_CATCH                                                                \
    }
    EV_UNARY(Evaluator_Negate, negate_inplace)
    EV_UNARY(Evaluator_TransformToNTT2, transform_to_ntt_inplace)
    EV_UNARY(Evaluator_TransformFromNTT, transform_from_ntt_inplace)
    EV_UNARY_POOL(Evaluator_Square, square_inplace)
    EV_UNARY_POOL(Evaluator_ModSwitchToNext1, mod_switch_to_next_inplace)
    EV_UNARY_POOL(Evaluator_RescaleToNext, rescale_to_next_inplace)
    EV_UNARY_POOL(Evaluator_ModReduceToNext, mod_reduce_to_next_inplace)

    // ------------------------------------------------------------------ plaintext operands, many-operand forms
#define EV_PLAIN(fn, call)                                                       \
    SHL_FUNC fn(void *thisptr, void *encrypted, void *plain, void *destination)  \
    {                                                                            \
        IfNullRet(thisptr, SHL_E_POINTER);                                       \
        IfNullRet(encrypted, SHL_E_POINTER);                                     \
        IfNullRet(plain, SHL_E_POINTER);                                         \
        IfNullRet(destination, SHL_E_POINTER);                                   \
        SHL_TRY                                                                  \
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());               \
        as<Evaluator>(thisptr)->call(prepare_dest(encrypted, destination), *as<Plaintext>(plain)); \
        SHL_CATCH                                                                \
    }
    EV_PLAIN(Evaluator_AddPlain, add_plain_inplace)
    EV_PLAIN(Evaluator_SubPlain, sub_plain_inplace)
    SHL_FUNC Evaluator_MultiplyPlain(void *thisptr, void *encrypted, void *plain, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->multiply_plain_inplace(prepare_dest(encrypted, destination), *as<Plaintext>(plain));
        SHL_CATCH
    }
    static Plaintext &prepare_plain_dest(void *plain, void *destination)
    {
        Plaintext *src = as<Plaintext>(plain), *dst = as<Plaintext>(destination);
        if (src != dst)
            *dst = *src;
        return *dst;
    }
    SHL_FUNC Evaluator_TransformToNTT1(void *thisptr, void *plain, uint64_t *parms_id, void *destination_ntt, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination_ntt, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->transform_to_ntt_inplace(prepare_plain_dest(plain, destination_ntt), parms_id);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ModSwitchToNext2(void *thisptr, void *plain, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->mod_switch_to_next_inplace(prepare_plain_dest(plain, destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ModSwitchTo2(void *thisptr, void *plain, uint64_t *parms_id, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->mod_switch_to_inplace(prepare_plain_dest(plain, destination), parms_id);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_AddMany(void *thisptr, uint64_t count, void **encrypteds, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypteds, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        std::vector<const Ciphertext *> v;
        for (uint64_t i = 0; i < count; i++)
            v.push_back(as<Ciphertext>(encrypteds[i]));
        as<Evaluator>(thisptr)->add_many(v, *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_MultiplyMany(void *thisptr, uint64_t count, void **encrypteds, void *relin_keys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypteds, SHL_E_POINTER);
        IfNullRet(relin_keys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        std::vector<const Ciphertext *> v;
        for (uint64_t i = 0; i < count; i++)
            v.push_back(as<Ciphertext>(encrypteds[i]));
        as<Evaluator>(thisptr)->multiply_many(v, *as<KSwitchKeys>(relin_keys), *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Exponentiate(void *thisptr, void *encrypted, uint64_t exponent, void *relin_keys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(relin_keys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->exponentiate_inplace(prepare_dest(encrypted, destination), exponent, *as<KSwitchKeys>(relin_keys));
        SHL_CATCH
    }

    SHL_FUNC Evaluator_Add(void *thisptr, void *encrypted1, void *encrypted2, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted1, SHL_E_POINTER);
        IfNullRet(encrypted2, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        auto ev = as<Evaluator>(thisptr);
        if (encrypted2 == destination && encrypted1 != destination)
            ev->add_inplace(*as<Ciphertext>(destination), *as<Ciphertext>(encrypted1)); // evaluator.h add(): commutes
        else
            ev->add_inplace(prepare_dest(encrypted1, destination), *as<Ciphertext>(encrypted2));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Sub(void *thisptr, void *encrypted1, void *encrypted2, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted1, SHL_E_POINTER);
        IfNullRet(encrypted2, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        auto ev = as<Evaluator>(thisptr);
        if (encrypted2 == destination && encrypted1 != destination)
        {
            // evaluator.h sub(): destination = e2 - e1, then negate
            ev->sub_inplace(*as<Ciphertext>(destination), *as<Ciphertext>(encrypted1));
            ev->negate_inplace(*as<Ciphertext>(destination));
        }
        else
            ev->sub_inplace(prepare_dest(encrypted1, destination), *as<Ciphertext>(encrypted2));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Multiply(void *thisptr, void *encrypted1, void *encrypted2, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted1, SHL_E_POINTER);
        IfNullRet(encrypted2, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        auto ev = as<Evaluator>(thisptr);
        if (encrypted1 == encrypted2 && encrypted1 == destination)
            ev->multiply_inplace(*as<Ciphertext>(destination), *as<Ciphertext>(destination));
        else
            ev->multiply(*as<Ciphertext>(encrypted1), *as<Ciphertext>(encrypted2), *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Relinearize(void *thisptr, void *encrypted, void *relinKeys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(relinKeys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->relinearize_inplace(prepare_dest(encrypted, destination), *as<KSwitchKeys>(relinKeys));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_SwitchKeyAccWords(void *thisptr, void *encrypted, uint64_t *words)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(words, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        *words = as<Evaluator>(thisptr)->switch_key_acc_words(*as<Ciphertext>(encrypted));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RelinearizePartial(void *thisptr, void *encrypted, void *relinKeys, uint64_t digit_first, uint64_t digit_count,
                                          uint64_t *device_acc)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(relinKeys, SHL_E_POINTER);
        IfNullRet(device_acc, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->relinearize_partial(*as<Ciphertext>(encrypted), *as<KSwitchKeys>(relinKeys), (unsigned)digit_first,
                                                    (unsigned)(digit_first + digit_count), device_acc);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RelinearizeFinish(void *thisptr, void *encrypted, uint64_t *device_acc, uint64_t parts)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(device_acc, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->relinearize_finish(*as<Ciphertext>(encrypted), device_acc, (unsigned)parts);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ApplyGaloisPartial(void *thisptr, void *encrypted, uint32_t galois_elt, void *galoisKeys, uint64_t digit_first,
                                          uint64_t digit_count, uint64_t *device_acc)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(device_acc, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->apply_galois_partial(*as<Ciphertext>(encrypted), galois_elt, *as<KSwitchKeys>(galoisKeys),
                                                     (unsigned)digit_first, (unsigned)(digit_first + digit_count), device_acc);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ApplyGaloisFinish(void *thisptr, void *encrypted, uint64_t *device_acc, uint64_t parts)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(device_acc, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->apply_galois_finish(*as<Ciphertext>(encrypted), device_acc, (unsigned)parts);
        SHL_CATCH
    }
    // ------------------------------------------------------------------ communicator + digit-parallel forms with the exchange
    // inside the library (sealhip.h section 1c; comm.h)
    SHL_FUNC Comm_GetUniqueId(uint8_t *id128)
    {
        IfNullRet(id128, SHL_E_POINTER);
        SHL_TRY
        Comm::unique_id(id128);
        SHL_CATCH
    }
    SHL_FUNC Comm_RcclAvailable(bool *available)
    {
        IfNullRet(available, SHL_E_POINTER);
        SHL_TRY
        *available = Comm::rccl_available();
        SHL_CATCH
    }
    SHL_FUNC Comm_Create(const uint8_t *id128, int nranks, int rank, void **comm)
    {
        IfNullRet(comm, SHL_E_POINTER);
        SHL_TRY
        *comm = new Comm(id128, nranks, rank);
        SHL_CATCH
    }
    SHL_FUNC Comm_Destroy(void *comm)
    {
        IfNullRet(comm, SHL_E_POINTER);
        delete as<Comm>(comm);
        return SHL_S_OK;
    }
    SHL_FUNC Comm_Info(void *comm, int *nranks, int *rank, bool *loopback)
    {
        IfNullRet(comm, SHL_E_POINTER);
        SHL_TRY
        if (nranks)
            *nranks = as<Comm>(comm)->size();
        if (rank)
            *rank = as<Comm>(comm)->rank();
        if (loopback)
            *loopback = as<Comm>(comm)->loopback();
        SHL_CATCH
    }
    SHL_FUNC Comm_DigitRange(void *comm, uint64_t digits, uint64_t *first, uint64_t *count)
    {
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(first, SHL_E_POINTER);
        IfNullRet(count, SHL_E_POINTER);
        SHL_TRY
        unsigned f, c;
        comm_split((unsigned)digits, (unsigned)as<Comm>(comm)->size(), (unsigned)as<Comm>(comm)->rank(), f, c);
        *first = f;
        *count = c;
        SHL_CATCH
    }
    SHL_FUNC Comm_AllReduceWords(void *comm, uint64_t *device_words, uint64_t count, void *hip_stream)
    {
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(device_words, SHL_E_POINTER);
        SHL_TRY
        as<Comm>(comm)->all_reduce_sum(device_words, (size_t)count, (hipStream_t)hip_stream);
        SHL_CATCH
    }
    SHL_FUNC Comm_BroadcastWords(void *comm, uint64_t *device_words, uint64_t count, int root, void *hip_stream)
    {
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(device_words, SHL_E_POINTER);
        SHL_TRY
        as<Comm>(comm)->broadcast(device_words, (size_t)count, root, (hipStream_t)hip_stream);
        SHL_CATCH
    }
    static Evaluator::KsExchange exchange_of(int how)
    {
        if (how != 0 && how != 1)
            throw std::invalid_argument("exchange: 0 = all-reduce, 1 = reduce-scatter + all-gather");
        return how ? Evaluator::KsExchange::reduce_scatter : Evaluator::KsExchange::all_reduce;
    }
    SHL_FUNC Evaluator_RelinearizeDigitParallel(void *thisptr, void *encrypted, void *relinKeys, void *comm, int exchange, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(relinKeys, SHL_E_POINTER);
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->relinearize_inplace(prepare_dest(encrypted, destination), *as<KSwitchKeys>(relinKeys), *as<Comm>(comm),
                                                    exchange_of(exchange));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ApplyGaloisDigitParallel(void *thisptr, void *encrypted, uint32_t galois_elt, void *galoisKeys, void *comm, int exchange,
                                                void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->apply_galois_inplace(prepare_dest(encrypted, destination), galois_elt, *as<KSwitchKeys>(galoisKeys),
                                                     *as<Comm>(comm), exchange_of(exchange));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RotateVectorDigitParallel(void *thisptr, void *encrypted, int steps, void *galoisKeys, void *comm, int exchange,
                                                 void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->rotate_vector_inplace(prepare_dest(encrypted, destination), steps, *as<KSwitchKeys>(galoisKeys), *as<Comm>(comm),
                                                      exchange_of(exchange));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_BroadcastKeyDigits(void *thisptr, void *kswitch_keys, uint64_t index, uint64_t *device_staging, void *comm, int root)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(kswitch_keys, SHL_E_POINTER);
        IfNullRet(device_staging, SHL_E_POINTER);
        IfNullRet(comm, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->broadcast_key_digits(*as<KSwitchKeys>(kswitch_keys), (size_t)index, device_staging, *as<Comm>(comm), root);
        SHL_CATCH
    }
    // the local phases of the reduce-scatter exchange (tests emulate the ranks in one process)
    SHL_FUNC Evaluator_SwitchKeySlots(void *thisptr, void *encrypted, uint64_t nranks, uint64_t *slots)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(slots, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        *slots = as<Evaluator>(thisptr)->switch_key_slots(*as<Ciphertext>(encrypted), (unsigned)nranks);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_SwitchKeyPackTargets(void *thisptr, void *encrypted, const uint64_t *device_acc, uint64_t nranks, uint64_t *device_send,
                                            uint64_t *device_special)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->switch_key_pack_targets(*as<Ciphertext>(encrypted), device_acc, (unsigned)nranks, device_send, device_special);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_SwitchKeyFinishOwned(void *thisptr, void *encrypted, const uint64_t *device_recv, const uint64_t *device_special,
                                            uint64_t nranks, uint64_t rank, uint64_t *device_own)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->switch_key_finish_owned(*as<Ciphertext>(encrypted), device_recv, device_special, (unsigned)nranks, (unsigned)rank,
                                                        device_own);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_SwitchKeyAddGathered(void *thisptr, void *encrypted, const uint64_t *device_all, uint64_t nranks)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->switch_key_add_gathered(*as<Ciphertext>(encrypted), device_all, (unsigned)nranks);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ModSwitchTo1(void *thisptr, void *encrypted, uint64_t *parms_id, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->mod_switch_to_inplace(prepare_dest(encrypted, destination), parms_id);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RescaleTo(void *thisptr, void *encrypted, uint64_t *parms_id, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->rescale_to_inplace(prepare_dest(encrypted, destination), parms_id);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ModReduceTo(void *thisptr, void *encrypted, uint64_t *parms_id, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->mod_reduce_to_inplace(prepare_dest(encrypted, destination), parms_id);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ApplyGalois(void *thisptr, void *encrypted, uint32_t galois_elt, void *galoisKeys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->apply_galois(*as<Ciphertext>(encrypted), galois_elt, *as<KSwitchKeys>(galoisKeys), *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RotateRows(void *thisptr, void *encrypted, int steps, void *galoisKeys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->rotate_rows(*as<Ciphertext>(encrypted), steps, *as<KSwitchKeys>(galoisKeys), *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RotateColumns(void *thisptr, void *encrypted, void *galois_keys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galois_keys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->rotate_columns(*as<Ciphertext>(encrypted), *as<KSwitchKeys>(galois_keys), *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RotateVector(void *thisptr, void *encrypted, int steps, void *galoisKeys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->rotate_vector(*as<Ciphertext>(encrypted), steps, *as<KSwitchKeys>(galoisKeys), *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ComplexConjugate(void *thisptr, void *encrypted, void *galoisKeys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->complex_conjugate(*as<Ciphertext>(encrypted), *as<KSwitchKeys>(galoisKeys), *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ContextUsingKeyswitching(void *thisptr, bool *using_keyswitching)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(using_keyswitching, SHL_E_POINTER);
        *using_keyswitching = as<Evaluator>(thisptr)->context().using_keyswitching();
        return SHL_S_OK;
    }

}
