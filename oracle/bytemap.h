/*
 * bytemap.h -- open-addressing hash map keyed by short byte strings, value = 2 x 64-bit words.
 * Stands in for the Rust std/ahash HashMap/HashSet the reference uses on this path
 * (metric/src/lib.rs:61-111 TxHashMap/TxHashSet; std::collections::HashMap in mark_dups.rs).
 * Oracle-internal (test infrastructure).
 */
#ifndef ORACLE_BYTEMAP_H
#define ORACLE_BYTEMAP_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BYTEMAP_KEY_MAX 32

typedef struct {
    uint8_t key[BYTEMAP_KEY_MAX];
    uint8_t len; /* 0 = empty slot (keys are never empty) */
    int64_t v0;
    int64_t v1;
} bytemap_slot;

typedef struct {
    bytemap_slot *slots;
    uint64_t cap; /* power of two */
    uint64_t size;
} bytemap;

static inline uint64_t bytemap_hash(const uint8_t *k, uint32_t len) {
    /* FNV-1a 64 followed by a murmur-style finaliser */
    uint64_t h = 0xcbf29ce484222325ULL;
    for (uint32_t i = 0; i < len; i++) {
        h ^= k[i];
        h *= 0x100000001b3ULL;
    }
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    return h;
}

static inline void bytemap_init(bytemap *m, uint64_t expected) {
    uint64_t cap = 16;
    while (cap < expected * 2 + 2) cap <<= 1;
    m->cap = cap;
    m->size = 0;
    m->slots = (bytemap_slot *)calloc(cap, sizeof(bytemap_slot));
}

static inline void bytemap_free(bytemap *m) {
    free(m->slots);
    m->slots = NULL;
    m->cap = m->size = 0;
}

static inline bytemap_slot *bytemap_find(const bytemap *m, const uint8_t *k, uint32_t len) {
    uint64_t i = bytemap_hash(k, len) & (m->cap - 1);
    for (;;) {
        bytemap_slot *s = &m->slots[i];
        if (s->len == 0) return NULL;
        if (s->len == len && memcmp(s->key, k, len) == 0) return s;
        i = (i + 1) & (m->cap - 1);
    }
}

static inline bytemap_slot *bytemap_insert_raw(bytemap *m, const uint8_t *k, uint32_t len, int *fresh);

static inline void bytemap_grow(bytemap *m) {
    bytemap old = *m;
    m->cap = old.cap * 2;
    m->size = 0;
    m->slots = (bytemap_slot *)calloc(m->cap, sizeof(bytemap_slot));
    for (uint64_t i = 0; i < old.cap; i++) {
        if (old.slots[i].len) {
            int fresh;
            bytemap_slot *s = bytemap_insert_raw(m, old.slots[i].key, old.slots[i].len, &fresh);
            s->v0 = old.slots[i].v0;
            s->v1 = old.slots[i].v1;
        }
    }
    free(old.slots);
}

static inline bytemap_slot *bytemap_insert_raw(bytemap *m, const uint8_t *k, uint32_t len, int *fresh) {
    uint64_t i = bytemap_hash(k, len) & (m->cap - 1);
    for (;;) {
        bytemap_slot *s = &m->slots[i];
        if (s->len == 0) {
            memcpy(s->key, k, len);
            s->len = (uint8_t)len;
            s->v0 = 0;
            s->v1 = 0;
            m->size++;
            *fresh = 1;
            return s;
        }
        if (s->len == len && memcmp(s->key, k, len) == 0) {
            *fresh = 0;
            return s;
        }
        i = (i + 1) & (m->cap - 1);
    }
}

/* entry(key).or_insert(0): returns the slot, *fresh = 1 when newly created (values zeroed) */
static inline bytemap_slot *bytemap_entry(bytemap *m, const uint8_t *k, uint32_t len, int *fresh) {
    if ((m->size + 1) * 2 > m->cap) bytemap_grow(m);
    return bytemap_insert_raw(m, k, len, fresh);
}

#endif
