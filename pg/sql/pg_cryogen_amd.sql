-- pg_cryogen_amd.sql -- regression script of the MI355X codec build of pg_cryogen (SURVEY.md row f-4).
--
-- Covers what the reference's own script covers (reference sql/pg_cryogen.sql:1-94): COPY into a cryo table
-- (multi_insert -> cryo_preserve -> cryo_compress), sequential / index / bitmap scans (cryo_read_data ->
-- cryo_decompress), a compression-method switch in the middle of a table (the method travels in each block's first page
-- header, reference cache.c:132-133), VACUUM, single-row INSERTs, a foreign key (tuple lock), a rescanned inner side of
-- a nested loop, and blocks whose compressed size needs a chain of several 8 KiB pages.  It adds the codec's own
-- parameters: every lz4 acceleration / zstd level class, and the GPU GUCs.
--
-- Every statement prints aggregates (count / sum / min / max / a checksum over md5 text), so the expected file is
-- derived by arithmetic, not by copying query output; it has NOT been produced by a server: this image has no
-- PostgreSQL (no pg_config, no server headers).  Run with:  make -C pg REF=... BATCH=1 installcheck REGRESS=pg_cryogen_amd
--
-- Two behaviours the reference's expected output pins as bugs are FIXED in the BATCH=1 build and this script expects
-- the fixed results (DESIGN.md section 7):
--   * a rescan of a cryo seq scan restarts it (reference expected/pg_cryogen.out:121-125 shows ONE row of a
--     three-row LIMIT because cryo_rescan never resets the block iterator);
--   * counting a table whose blocks span several pages works (reference expected/pg_cryogen.out:166 ends in
--     "iternal error; block 3 is not the part of seqscan iterator").

CREATE EXTENSION pg_cryogen;
SHOW pg_cryogen.gpu_device;
SHOW pg_cryogen.gpu_count;
SHOW pg_cryogen.gpu_pool_mb;
SHOW pg_cryogen.gpu_workspace_keep_mb;

-- 1. COPY, default method (zstd level 1)
CREATE TABLE cold_events (id int4 NOT NULL, tag text) USING pg_cryogen;
COPY (SELECT g, md5(g::text) FROM generate_series(1, 2000) g) TO '/tmp/pg_cryogen_amd_events.csv' WITH csv;
COPY cold_events FROM '/tmp/pg_cryogen_amd_events.csv' WITH csv;
SELECT count(*), sum(id), min(id), max(id) FROM cold_events;
SELECT count(*) FROM cold_events WHERE tag = md5(id::text);

-- 2. index scan, then BRIN bitmap scan
CREATE INDEX cold_events_id ON cold_events USING btree (id);
ANALYZE cold_events;
SET enable_seqscan = off;
EXPLAIN (COSTS OFF) SELECT tag FROM cold_events WHERE id = 1234;
SELECT tag = md5('1234') AS same FROM cold_events WHERE id = 1234;
DROP INDEX cold_events_id;
CREATE INDEX cold_events_id ON cold_events USING brin (id);
EXPLAIN (COSTS OFF) SELECT tag FROM cold_events WHERE id = 1234;
SELECT tag = md5('1234') AS same FROM cold_events WHERE id = 1234;
SET enable_seqscan = on;

-- 3. switch the method in the middle of the table: old blocks stay zstd, new ones are lz4, both decode
SET pg_cryogen.compression_method = 'lz4';
COPY cold_events FROM '/tmp/pg_cryogen_amd_events.csv' WITH csv;
SELECT count(*), sum(id) FROM cold_events;
SELECT count(*) FROM cold_events WHERE tag = md5(id::text);

-- 4. every parameter class of both codecs (bytes are liblz4 1.9.3 / libzstd 1.4.8's; only the round trip shows here)
SET pg_cryogen.lz4_acceleration = 0;
COPY cold_events FROM '/tmp/pg_cryogen_amd_events.csv' WITH csv;
SET pg_cryogen.lz4_acceleration = 50;
COPY cold_events FROM '/tmp/pg_cryogen_amd_events.csv' WITH csv;
SET pg_cryogen.compression_method = 'zstd';
SET pg_cryogen.zstd_compression_level = -5;
COPY cold_events FROM '/tmp/pg_cryogen_amd_events.csv' WITH csv;
SET pg_cryogen.zstd_compression_level = 3;
COPY cold_events FROM '/tmp/pg_cryogen_amd_events.csv' WITH csv;
SET pg_cryogen.zstd_compression_level = 9;
COPY cold_events FROM '/tmp/pg_cryogen_amd_events.csv' WITH csv;
SET pg_cryogen.zstd_compression_level = 22;
COPY cold_events FROM '/tmp/pg_cryogen_amd_events.csv' WITH csv;
SELECT count(*), sum(id), count(DISTINCT tag) FROM cold_events;
SELECT count(*) FROM cold_events WHERE tag = md5(id::text);
RESET pg_cryogen.zstd_compression_level;
RESET pg_cryogen.lz4_acceleration;
RESET pg_cryogen.compression_method;

-- 5. VACUUM keeps the table readable
VACUUM cold_events;
SELECT count(*) FROM cold_events;

-- 6. single-row inserts after TRUNCATE (tuple_insert, one block per statement end)
TRUNCATE cold_events;
INSERT INTO cold_events SELECT g, md5(g::text) FROM generate_series(1, 2000) g;
SELECT count(*), sum(id) FROM cold_events;

-- 7. foreign key: the referencing insert locks rows of the cryo table
DROP INDEX cold_events_id;
CREATE UNIQUE INDEX cold_events_id ON cold_events USING btree (id);
CREATE TABLE event_notes (note_id serial, event_id int4 REFERENCES cold_events (id));
INSERT INTO event_notes VALUES (1, 70), (2, 700), (3, 1700);
SELECT count(*), sum(e.id) FROM cold_events e JOIN event_notes n ON n.event_id = e.id;
DROP TABLE event_notes;
DROP INDEX cold_events_id;

-- 8. rescan: the cryo table as the inner side of a nested loop is scanned once per outer row (FIXED behaviour)
CREATE TABLE outer_side AS SELECT g AS id, g * 3 AS w FROM generate_series(1, 40) g;
SET enable_hashjoin = off;
SET enable_mergejoin = off;
SET enable_material = off;
EXPLAIN (COSTS OFF) SELECT * FROM cold_events JOIN outer_side USING (id);
SELECT count(*), sum(w) FROM cold_events JOIN outer_side USING (id);
RESET enable_hashjoin;
RESET enable_mergejoin;
RESET enable_material;

-- 9. blocks that need a chain of several pages: wide, poorly compressible rows (FIXED behaviour: the count works)
CREATE TABLE wide_docs (doc jsonb) USING pg_cryogen;
BEGIN;
INSERT INTO wide_docs VALUES ('{"rolled": "back"}');
ROLLBACK;
INSERT INTO wide_docs
SELECT jsonb_build_object('id', g) ||
       (SELECT jsonb_object_agg('k' || j, md5(g::text || '/' || j::text)) FROM generate_series(1, 26) j)
FROM generate_series(1, 600) g;
SELECT count(*), sum((doc->>'id')::int) FROM wide_docs;
SELECT count(*) FROM wide_docs WHERE doc->>'k7' = md5((doc->>'id') || '/7');

-- 10. a re-scan served from the device-resident pool returns the same rows
SET pg_cryogen.gpu_pool_mb = 64;
SELECT count(*), sum((doc->>'id')::int) FROM wide_docs;
SELECT count(*), sum((doc->>'id')::int) FROM wide_docs;
RESET pg_cryogen.gpu_pool_mb;

DROP TABLE wide_docs;
DROP TABLE outer_side;
DROP TABLE cold_events;
DROP EXTENSION pg_cryogen;
