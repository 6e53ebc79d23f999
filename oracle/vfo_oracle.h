/* oracle/vfo_oracle.h -- TEST INFRASTRUCTURE (see vfo_oracle.c). */
#ifndef VFO_ORACLE_H
#define VFO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_vfo orc_vfo;

/* ---- primitives ------------------------------------------------------- */
long orc_osc_table(double fs, double f, float *table_iq);
void orc_osc_sequence(double fs, double f, long n_ticks, float *out_iq);
int orc_low_pass(double gain, double fs, double fc, double tw, float *taps, int max);
void orc_hilbert_taps(int len, int fs, float *taps);
void orc_dc_correct(float *iq, int n_complex, float state[2]);
void orc_u8_to_float(const unsigned char *bytes, int n, float *out);
short orc_double_to_short(double d);
signed char orc_float_to_schar(float f);

/* ---- one VFO node ----------------------------------------------------- */
orc_vfo *orc_vfo_new(void);
void orc_vfo_free(orc_vfo *v);
void orc_vfo_set_fs(orc_vfo *v, int fs);
void orc_vfo_set_decimation_count(orc_vfo *v, int count);
void orc_vfo_set_mixer_freq(orc_vfo *v, double f);
void orc_vfo_set_demod_usb(orc_vfo *v, int usb);
void orc_vfo_set_filter_bandwidth(orc_vfo *v, double bw);
void orc_vfo_set_gain(orc_vfo *v, float g);
void orc_vfo_set_compression_style(orc_vfo *v, int st);
void orc_vfo_set_scale_comp(orc_vfo *v, int s);
void orc_vfo_set_topic(orc_vfo *v, const char *topic);
int orc_vfo_init(orc_vfo *v, int samples_per_buffer, int late_decimate);
void orc_vfo_add_child(orc_vfo *parent, orc_vfo *child);
void orc_vfo_process(orc_vfo *v, const float *iq, int n_complex);
void orc_process_roots(orc_vfo **roots, int n_roots, const float *iq, int n_complex, int frames,
                       int threads);

int orc_vfo_decimate_count(const orc_vfo *v);
unsigned orc_vfo_output_rate(const orc_vfo *v);
int orc_vfo_get_stream(const orc_vfo *v, int stage, float *out_iq, int max_complex);
int orc_vfo_get_usb(const orc_vfo *v, short *out, int max);
int orc_vfo_get_usb_prequant(const orc_vfo *v, double *out, int max);
int orc_vfo_get_iq(const orc_vfo *v, signed char *out, int max);
int orc_vfo_get_fir_usb_taps(const orc_vfo *v, float *out, int max);
int orc_vfo_get_fir_dec_taps(const orc_vfo *v, float *out, int max);
int orc_vfo_get_hilbert_taps(const orc_vfo *v, float *out, int max);
int orc_vfo_get_publish(const orc_vfo *v, char topic5[5], unsigned *rate, const unsigned char **payload,
                        unsigned *len);

#ifdef __cplusplus
}
#endif
#endif
