/*
 * Type-generic body of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 * Included twice by stft_oracle.c with
 *   T      element type (float / double)
 *   SFX    real suffix (s / d),  CFX complex suffix (c / z)
 *   ROUND  roundf / round,  SQRT sqrtf / sqrt,  LOG logf / log,  EPS FLT_EPSILON / DBL_EPSILON
 * Every function restates one reference function; the citation is on each.
 */

#define CAT_(a,b) a##b
#define CAT(a,b) CAT_(a,b)
#define FN(name) CAT(CAT(orc_,name),CAT(_,SFX))
#define FNC(name) CAT(CAT(orc_,name),CAT(_,CFX))
#define STR_(x) #x
#define STR(x) STR_(x)

/* ---- half-complex real DFT of one zero-padded frame -------------------------------------
 * Replaces fftw(f)_execute of an FFTW_R2HC r2r plan (reference c/stft.c:53,77; FFTW3, unpinned
 * version, not in the reference tree).  Published definition: Y_k = sum_j X_j e^{-2 pi i jk/n},
 * unnormalised, output order r0..r_{n/2}, i_{(n+1)/2-1}..i_1.
 * mode 0: long-double evaluation rounded once to T (error below FFTW's own rounding error);
 * mode 1: arithmetic in T (the timed CPU baseline);
 * mode 2: FFTW itself, where the benchmark found libfftw3(f) on the host and handed its three entry points
 *         over with orc_set_fftw_s/_d: an FFTW_R2HC r2r plan with FFTW_ESTIMATE on this (in, out) pair, made
 *         when the pair changes, i.e. once per call of stft, as the reference does (c/stft.c:53,77,158).
 *         FFTW's planner is not thread-safe: plans are made and destroyed under a lock. */
typedef void *(*FN(fftw_plan_fn))(int, T *, T *, int, unsigned);
typedef void (*FN(fftw_exec_fn))(void *);
static FN(fftw_plan_fn) FN(fftw_plan) = NULL;
static FN(fftw_exec_fn) FN(fftw_exec) = NULL, FN(fftw_destroy) = NULL;
void FN(set_fftw)(void *plan, void *exec, void *destroy)
{
    FN(fftw_plan) = (FN(fftw_plan_fn))plan; FN(fftw_exec) = (FN(fftw_exec_fn))exec; FN(fftw_destroy) = (FN(fftw_exec_fn))destroy;
}
static void FN(r2hc)(T *out, const T *in, size_t n, int mode)
{
    if (mode==2 && FN(fftw_plan))
    {
        static __thread void *plan = NULL; static __thread const T *pin = NULL; static __thread T *pout = NULL; static __thread size_t pn = 0u;
        if (plan==NULL || pin!=in || pout!=out || pn!=n)
        {
            while (__atomic_exchange_n(&orc_fftw_lock,1,__ATOMIC_ACQUIRE)) { }
            if (plan) { FN(fftw_destroy)(plan); }
            plan = FN(fftw_plan)((int)n,(T*)in,out,0 /* FFTW_R2HC */,1u<<6 /* FFTW_ESTIMATE */);
            __atomic_store_n(&orc_fftw_lock,0,__ATOMIC_RELEASE);
            pin = in; pout = out; pn = n;
        }
        if (plan) { FN(fftw_exec)(plan); return; }
    }
    if (mode==1 && n>=4u && !(n&(n-1u)))
    {
        const size_t m = n/2u;
        static __thread T *z = NULL, *tw = NULL; static __thread size_t cap = 0u, twn = 0u;
        if (cap<n) { free(z); z = (T*)malloc(2u*n*sizeof(T)); cap = n; }
        if (twn!=n)
        {
            free(tw); tw = (T*)malloc(2u*n*sizeof(T)); twn = n;
            for (size_t j=0u; j<n; ++j)
            {
                const long double a = -2.0L*PI_L*(long double)j/(long double)n;
                tw[2u*j] = (T)cosl(a); tw[2u*j+1u] = (T)sinl(a);
            }
        }
        /* even/odd packing: z_j = x_{2j} + i x_{2j+1}, bit-reversed load */
        unsigned lg = 0u; while ((1u<<lg)<m) { ++lg; }
        for (size_t j=0u; j<m; ++j)
        {
            size_t r = 0u; for (unsigned b=0u; b<lg; ++b) { r |= ((j>>b)&1u)<<(lg-1u-b); }
            z[2u*r] = in[2u*j]; z[2u*r+1u] = in[2u*j+1u];
        }
        for (size_t h=1u; h<m; h*=2u)
        {
            const size_t ts = n/(4u*h)*2u;   /* twiddle stride in the n-point table: W_{2h}^j = W_n^{j n/(2h)} */
            for (size_t i=0u; i<m; i+=2u*h)
            {
                for (size_t j=0u; j<h; ++j)
                {
                    const T wr = tw[2u*(j*ts)], wi = tw[2u*(j*ts)+1u];
                    T *a = z+2u*(i+j), *b = z+2u*(i+j+h);
                    const T tr = b[0]*wr - b[1]*wi, ti = b[0]*wi + b[1]*wr;
                    b[0] = a[0]-tr; b[1] = a[1]-ti; a[0] += tr; a[1] += ti;
                }
            }
        }
        /* split: X_k = E_k + W_n^k O_k,  E = (Z_k + conj Z_{m-k})/2,  O = -i (Z_k - conj Z_{m-k})/2 */
        out[0] = z[0]+z[1]; out[m] = z[0]-z[1];
        for (size_t k=1u; k<m; ++k)
        {
            const T ar = z[2u*k], ai = z[2u*k+1u], br = z[2u*(m-k)], bi = -z[2u*(m-k)+1u];
            const T er = (T)0.5*(ar+br), ei = (T)0.5*(ai+bi);
            const T pr = (T)0.5*(ar-br), pi = (T)0.5*(ai-bi);   /* (Z_k - conj Z_{m-k})/2 */
            const T or_ = pi, oi = -pr;                          /* times -i */
            const T wr = tw[2u*k], wi = tw[2u*k+1u];
            const T xr = er + or_*wr - oi*wi, xi = ei + or_*wi + oi*wr;
            out[k] = xr; if (k<m) { out[n-k] = xi; }
        }
        return;
    }

    /* mode 0 (or sizes the T-precision path does not cover) */
    long double *re = (long double*)malloc(2u*n*sizeof(long double)), *im = re+n;
    if (n>=2u && !(n&(n-1u)))
    {
        unsigned lg = 0u; while (((size_t)1u<<lg)<n) { ++lg; }
        for (size_t j=0u; j<n; ++j)
        {
            size_t r = 0u; for (unsigned b=0u; b<lg; ++b) { r |= ((j>>b)&1u)<<(lg-1u-b); }
            re[r] = (long double)in[j]; im[r] = 0.0L;
        }
        for (size_t h=1u; h<n; h*=2u)
        {
            for (size_t j=0u; j<h; ++j)
            {
                const long double a = -PI_L*(long double)j/(long double)h;
                const long double wr = cosl(a), wi = sinl(a);
                for (size_t i=j; i<n; i+=2u*h)
                {
                    const long double tr = re[i+h]*wr - im[i+h]*wi, ti = re[i+h]*wi + im[i+h]*wr;
                    re[i+h] = re[i]-tr; im[i+h] = im[i]-ti; re[i] += tr; im[i] += ti;
                }
            }
        }
        out[0] = (T)re[0];
        for (size_t k=1u; k<=n/2u; ++k) { out[k] = (T)re[k]; if (k<(n+1u)/2u) { out[n-k] = (T)im[k]; } }
    }
    else
    {
        long double *c = (long double*)malloc(2u*n*sizeof(long double)), *s = c+n;
        for (size_t j=0u; j<n; ++j)
        {
            const long double a = 2.0L*PI_L*(long double)j/(long double)n;
            c[j] = cosl(a); s[j] = sinl(a);
        }
        for (size_t k=0u; k<=n/2u; ++k)
        {
            long double sr = 0.0L, si = 0.0L;
            for (size_t t=0u; t<n; ++t) { const size_t a = (k*t)%n; sr += (long double)in[t]*c[a]; si -= (long double)in[t]*s[a]; }
            out[k] = (T)sr; if (k>0u && k<(n+1u)/2u) { out[n-k] = (T)si; }
        }
        free(c);
    }
    free(re);
}


/* ---- power + amp/lg of one half-complex frame; reference c/stft.c:80-95 (loops kept literally,
 * so the pairing for odd nfft is the reference's) */
static void FN(power_row)(T *Y, const T *Yw, size_t nfft, int amp, int lg)
{
    const size_t F = nfft/2u + 1u;
    for (size_t f=0u; f<F; ++f) { Y[f] = Yw[f]*Yw[f]; }
    for (size_t j=0u; j+2u<F; ++j) { Y[F-2u-j] += Yw[F+j]*Yw[F+j]; }
    if (amp)
    {
        if (lg) { for (size_t f=0u; f<F; ++f) { Y[f] = LOG(SQRT(Y[f])); } }
        else { for (size_t f=0u; f<F; ++f) { Y[f] = SQRT(Y[f]); } }
    }
    else if (lg) { for (size_t f=0u; f<F; ++f) { Y[f] = LOG(Y[f]); } }
}

/* mean removal over the first L samples, summed from the last sample down; c/stft.c:67-73 */
static void FN(zero_mean)(T *Xw, size_t L)
{
    T mn = (T)0;
    for (size_t l=L; l>0u; --l) { mn += Xw[l-1u]; }
    mn /= (T)L;
    for (size_t l=0u; l<L; ++l) { Xw[l] -= mn; }
}

static long long FN(fold)(long long n, long long N)
{
    /* c/stft.c:113 */
    while (n<0 || n>=N) { n = (n<0) ? -n-1 : (n<N) ? n : 2*N-1-n; }
    return n;
}


/* ---- stft_s / stft_d : reference c/stft.c:33-162, :165-301 ---- */
int FN(stft)(T *Y, const T *X1, const T *X2, const size_t N, const size_t L, const size_t nfft, const size_t stp,
             const int snip_edges, const int mn0, const int amp, const int lg, const int mode)
{
    if (L<1u) { fprintf(stderr,"error in stft_" STR(SFX) ": L must be positive\n"); return 1; }
    if (stp<1u) { fprintf(stderr,"error in stft_" STR(SFX) ": stp must be positive\n"); return 1; }
    if (snip_edges && L>N) { fprintf(stderr,"error in stft_" STR(SFX) ": L must be < N if snip_edges\n"); return 1; }
    /* the reference has undefined behaviour below these two bounds (buffer overrun / F-2 underflow) */
    if (nfft<L) { fprintf(stderr,"error in stft_" STR(SFX) ": nfft must be >= L\n"); return 1; }
    if (nfft<2u) { fprintf(stderr,"error in stft_" STR(SFX) ": nfft must be >= 2\n"); return 1; }

    const size_t W = (snip_edges) ? 1u+(N-L)/stp : (N+stp/2u)/stp;          /* c/stft.c:40 */
    if (W==0u) { return 0; }
    const size_t F = nfft/2u + 1u;
    T *Xw = (T*)calloc(2u*nfft,sizeof(T)), *Yw = Xw+nfft;
    long long ss = snip_edges ? 0 : (long long)(stp/2u) - (long long)(L/2u);    /* c/stft.c:100-101 */
    for (size_t w=0u; w<W; ++w, ss+=(long long)stp, Y+=F)
    {
        if (snip_edges || (ss>=0 && ss<=(long long)N-(long long)L))
        {
            for (size_t l=0u; l<L; ++l) { Xw[l] = X1[(size_t)ss+l]*X2[l]; }          /* :63, :121 */
        }
        else
        {
            for (size_t l=0u; l<L; ++l) { Xw[l] = X1[FN(fold)(ss+(long long)l,(long long)N)]*X2[l]; }  /* :109-116 */
        }
        if (mn0) { FN(zero_mean)(Xw,L); }
        FN(r2hc)(Yw,Xw,nfft,mode);
        FN(power_row)(Y,Yw,nfft,amp,lg);
    }
    free(Xw);
    return 0;
}


/* ---- frame start table of the _flt functions: serial accumulation of the centre in T, rounding half
 * away from zero; c/stft_flt.c:57-61,91-92 ; c/frame_univar_flt.c:44-46,52 ---- */
void FN(flt_starts)(int *ss_out, const size_t L, const size_t W, const T c0, const T stp)
{
    T cc = c0;
    for (size_t w=0u; w<W; ++w)
    {
        ss_out[w] = (int)ROUND(cc) - (int)(L/2u);
        cc += stp;
    }
}


/* ---- stft_flt_s / _d : reference c/stft_flt.c:33-129, :132-228.
 * intended==0 walks the reference's X1 pointer literally (p is X1's offset), which reproduces the
 * reference's actual output, including that p is never advanced to the first non-negative start;
 * intended==1 gives the framing of c/window_univar_flt.c:49-87 (zeros outside [0,N-1]). */
int FN(stft_flt)(T *Y, const T *X1, const T *X2, const size_t N, const size_t L, const size_t W, const size_t nfft,
                 const T c0, const T stp, const int mn0, const int amp, const int lg, const int intended, const int mode)
{
    if (W>N) { fprintf(stderr,"error in stft_flt_" STR(SFX) ": W must be <= N (length X)\n"); return 1; }
    if (L>N) { fprintf(stderr,"error in stft_flt_" STR(SFX) ": L must be <= N (length X)\n"); return 1; }
    if (L<1u) { fprintf(stderr,"error in stft_flt_" STR(SFX) ": L must be positive\n"); return 1; }
    if (N<1u) { fprintf(stderr,"error in stft_flt_" STR(SFX) ": N (length X) must be positive\n"); return 1; }
    if (c0>(T)(N-1u)) { fprintf(stderr,"error in stft_flt_" STR(SFX) ": c0 (center samp of 1st frame) must be < N (length X)\n"); return 1; }
    if (stp<EPS) { fprintf(stderr,"error in stft_flt_" STR(SFX) ": stp (step size) must be positive\n"); return 1; }
    if (nfft<L) { fprintf(stderr,"error in stft_flt_" STR(SFX) ": nfft must be >= L\n"); return 1; }
    if (nfft<2u) { fprintf(stderr,"error in stft_flt_" STR(SFX) ": nfft must be >= 2\n"); return 1; }
    if (W==0u) { return 0; }

    const size_t F = nfft/2u + 1u;
    T *Xw = (T*)calloc(2u*nfft,sizeof(T)), *Yw = Xw+nfft;
    const int Lpre = (int)(L/2u), Lpost = (int)L-Lpre-1;
    T cc = c0;
    int cs = (int)ROUND(c0), ss = cs-Lpre, es = cs+Lpost;
    long long p = 0;                                            /* offset of the reference's X1 pointer */
    for (size_t w=0u; w<W; ++w, Y+=F)
    {
        if (es<0 || ss>(int)N-1) { for (size_t l=0u; l<L; ++l) { Xw[l] = (T)0; } }
        else if (intended)
        {
            for (size_t l=0u; l<L; ++l)
            {
                const long long n = (long long)ss + (long long)l;
                Xw[l] = (n>=0 && n<(long long)N) ? X1[n]*X2[l] : (T)0;
            }
        }
        else if (ss<0)
        {
            size_t l = 0u;
            for (; l<(size_t)(-ss); ++l) { Xw[l] = (T)0; }
            for (long long q=p; l<L; ++l, ++q) { Xw[l] = X1[q]*X2[l]; }          /* :75-76: pointer restored */
        }
        else if (es<(int)N)
        {
            for (size_t l=0u; l<L; ++l) { Xw[l] = X1[p+(long long)l]*X2[l]; }
            p += (long long)((int)ROUND(cc+stp) - cs);                               /* :80-81 */
        }
        else
        {
            size_t l = 0u;
            for (; l<N-(size_t)ss; ++l) { Xw[l] = X1[p+(long long)l]*X2[l]; }
            for (; l<L; ++l) { Xw[l] = (T)0; }
            p += (long long)((int)ROUND(cc+stp) - cs);                               /* :85-87 */
        }
        cc += stp; cs = (int)ROUND(cc); ss = cs-Lpre; es = cs+Lpost;
        if (mn0) { FN(zero_mean)(Xw,L); }
        FN(r2hc)(Yw,Xw,nfft,mode);
        FN(power_row)(Y,Yw,nfft,amp,lg);
    }
    free(Xw);
    return 0;
}


/* ---- frame_univar_{s,d,c,z} : reference c/frame_univar.c:41-266. E = Ts per sample (1 real, 2 complex) ---- */
static int CAT(orc_frame_univar_impl_,SFX)(T *Y, const T *X, const size_t N, const size_t L, const size_t stp,
                                           const int snip_edges, const size_t E, const char *name)
{
    if (L<1u) { fprintf(stderr,"error in %s: L must be positive\n",name); return 1; }
    if (stp<1u) { fprintf(stderr,"error in %s: stp must be positive\n",name); return 1; }
    if (snip_edges && L>N) { fprintf(stderr,"error in %s: L must be < N if snip_edges\n",name); return 1; }
    const size_t W = (snip_edges) ? 1u+(N-L)/stp : (N+stp/2u)/stp;
    long long ss = snip_edges ? 0 : (long long)(stp/2u) - (long long)(L/2u);
    for (size_t w=0u; w<W; ++w, ss+=(long long)stp)
    {
        for (size_t l=0u; l<L; ++l)
        {
            const long long n = snip_edges ? ss+(long long)l : FN(fold)(ss+(long long)l,(long long)N);
            for (size_t e=0u; e<E; ++e) { *Y++ = X[(size_t)n*E+e]; }
        }
    }
    return 0;
}
int FN(frame_univar)(T *Y, const T *X, const size_t N, const size_t L, const size_t stp, const int snip_edges)
{ return CAT(orc_frame_univar_impl_,SFX)(Y,X,N,L,stp,snip_edges,1u,"frame_univar_" STR(SFX)); }
int FNC(frame_univar)(T *Y, const T *X, const size_t N, const size_t L, const size_t stp, const int snip_edges)
{ return CAT(orc_frame_univar_impl_,SFX)(Y,X,N,L,stp,snip_edges,2u,"frame_univar_" STR(CFX)); }


/* ---- frame_univar_flt_{s,d,c,z} : reference c/frame_univar_flt.c:29-286 ---- */
static int CAT(orc_frame_univar_flt_impl_,SFX)(T *Y, const T *X, const size_t N, const size_t L, const size_t W,
                                               const T c0, const T stp, const size_t E, const char *name)
{
    if (W>N) { fprintf(stderr,"error in %s: W must be <= N (length X)\n",name); return 1; }
    if (L>N) { fprintf(stderr,"error in %s: L must be <= N (length X)\n",name); return 1; }
    if (L<1u) { fprintf(stderr,"error in %s: L must be positive\n",name); return 1; }
    if (N<1u) { fprintf(stderr,"error in %s: N (length X) must be positive\n",name); return 1; }
    if (c0>(T)(N-1u)) { fprintf(stderr,"error in %s: c0 (center samp of 1st frame) must be < N (length X)\n",name); return 1; }
    if (stp<EPS) { fprintf(stderr,"error in %s: stp (step size) must be positive\n",name); return 1; }
    T cc = c0;
    for (size_t w=0u; w<W; ++w, cc+=stp)
    {
        const long long ss = (long long)((int)ROUND(cc) - (int)(L/2u));
        for (size_t l=0u; l<L; ++l)
        {
            const long long n = ss + (long long)l;
            for (size_t e=0u; e<E; ++e) { *Y++ = (n>=0 && n<(long long)N) ? X[(size_t)n*E+e] : (T)0; }
        }
    }
    return 0;
}
int FN(frame_univar_flt)(T *Y, const T *X, const size_t N, const size_t L, const size_t W, const T c0, const T stp)
{ return CAT(orc_frame_univar_flt_impl_,SFX)(Y,X,N,L,W,c0,stp,1u,"frame_univar_flt_" STR(SFX)); }
int FNC(frame_univar_flt)(T *Y, const T *X, const size_t N, const size_t L, const size_t W, const T c0, const T stp)
{ return CAT(orc_frame_univar_flt_impl_,SFX)(Y,X,N,L,W,c0,stp,2u,"frame_univar_flt_" STR(CFX)); }


/* ---- apply_win_{s,d,c,z} (the _inplace_ forms are the same map with Y==X1) : reference
 * c/apply_win.c:26-671.  Y = X1 .* X2 with X2 broadcast along dim of an R x C x S x H tensor.
 * The reference's strided branch never rewinds X2 and (row-major complex, not in place) uses
 * real strides; the map below is the one its contiguous branches and its 2-D cases compute. */
static int CAT(orc_apply_win_impl_,SFX)(T *Y, const T *X1, const T *X2, const size_t R, const size_t C, const size_t S,
                                        const size_t H, const size_t L, const size_t dim, const int iscolmajor,
                                        const int cplx, const char *name)
{
    if (dim>3u) { fprintf(stderr,"error in %s: dim must be in [0 3]\n",name); return 1; }
    if (dim==0u && R!=L) { fprintf(stderr,"error in %s: L (winlength) must equal R (nrows X) for dim=0\n",name); return 1; }
    if (dim==1u && C!=L) { fprintf(stderr,"error in %s: L (winlength) must equal C (ncols X) for dim=1\n",name); return 1; }
    if (dim==2u && S!=L) { fprintf(stderr,"error in %s: L (winlength) must equal S (nslices X) for dim=2\n",name); return 1; }
    if (dim==3u && H!=L) { fprintf(stderr,"error in %s: L (winlength) must equal H (nhyperslices X) for dim=3\n",name); return 1; }
    const size_t sz[4] = {R,C,S,H};
    size_t st[4];
    if (iscolmajor) { st[0]=1u; st[1]=R; st[2]=R*C; st[3]=R*C*S; }
    else { st[3]=1u; st[2]=H; st[1]=H*S; st[0]=H*S*C; }
    const size_t N = R*C*S*H;
    for (size_t n=0u; n<N; ++n)
    {
        const size_t l = (n/st[dim])%sz[dim];
        if (cplx)
        {
            const T ar = X1[2u*n], ai = X1[2u*n+1u], br = X2[2u*l], bi = X2[2u*l+1u];
            Y[2u*n] = ar*br - ai*bi; Y[2u*n+1u] = ar*bi + ai*br;
        }
        else { Y[n] = X1[n]*X2[l]; }
    }
    return 0;
}
int FN(apply_win)(T *Y, const T *X1, const T *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor)
{ return CAT(orc_apply_win_impl_,SFX)(Y,X1,X2,R,C,S,H,L,dim,iscolmajor,0,"apply_win_" STR(SFX)); }
int FNC(apply_win)(T *Y, const T *X1, const T *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor)
{ return CAT(orc_apply_win_impl_,SFX)(Y,X1,X2,R,C,S,H,L,dim,iscolmajor,1,"apply_win_" STR(CFX)); }


/* ---- fft_fftw_r2hc_{s,d} : reference c/fft.fftw.r2hc.c:24-109, :112-198.  Real FFT along dim with
 * zero-pad to nfft; Y has Ly = nfft/2+1 interleaved-complex entries along dim, imag(DC) = 0 and, for
 * even nfft, imag(Nyquist) = 0 (:74-77).  sc scales every output by 1/sqrt(2 nfft) (:101-105; the
 * reference applies it from an advanced pointer, which is not reproduced). */
int FN(fft_r2hc)(T *Y, const T *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor,
                 const size_t dim, const size_t nfft, const int sc, const int mode)
{
    if (dim>3u) { fprintf(stderr,"error in fft_fftw_r2hc_" STR(SFX) ": dim must be in [0 3]\n"); return 1; }
    const size_t sz[4] = {R,C,S,H};
    const size_t N = R*C*S*H, Lx = sz[dim], Ly = nfft/2u + 1u;
    if (nfft<Lx) { fprintf(stderr,"error in fft_fftw_r2hc_" STR(SFX) ": nfft must be >= L (vec length)\n"); return 1; }
    if (nfft==0u || N==0u) { return 0; }
    /* K = stride along dim, V = number of vectors; vector v = (outer o, inner i), i<K */
    size_t K = 1u;
    if (iscolmajor) { for (size_t d=0u; d<dim; ++d) { K *= sz[d]; } }
    else { for (size_t d=3u; d>dim; --d) { K *= sz[d]; } }
    const size_t V = N/Lx;
    T *xin = (T*)calloc(2u*nfft+2u,sizeof(T)), *hc = xin+nfft;
    const T s = (T)(1.0/sqrt((double)(2u*nfft)));
    for (size_t v=0u; v<V; ++v)
    {
        const size_t o = v/K, i = v%K;
        const T *x = X + o*K*Lx + i;
        T *y = Y + 2u*(o*K*Ly + i);
        for (size_t l=0u; l<Lx; ++l) { xin[l] = x[l*K]; }
        if (nfft==1u) { hc[0] = xin[0]; } else { FN(r2hc)(hc,xin,nfft,mode); }
        for (size_t k=0u; k<Ly; ++k)
        {
            T re = hc[k], im = (k>0u && k<(nfft+1u)/2u) ? hc[nfft-k] : (T)0;
            if (sc) { re *= s; im *= s; }
            y[2u*k*K] = re; y[2u*k*K+1u] = im;
        }
    }
    free(xin);
    return 0;
}


/* ---- fft_power_c / _z : reference c/fft_power.c:17-26, :29-34 ---- */
int FNC(fft_power)(T *Y, const T *X, const size_t N)
{
    for (size_t n=0u; n<N; ++n) { T y = X[2u*n]*X[2u*n]; y += X[2u*n+1u]*X[2u*n+1u]; Y[n] = y; }
    return 0;
}


/* ---- pow_compress_s / _d : reference c/pow_compress.c:23-50, :53-80 ---- */
int FN(pow_compress)(T *Y, const T *X, const size_t N, const T p, const T preg)
{
    if (p<(T)0 || p>(T)1) { fprintf(stderr,"error in %s: pow must be in [0.0 1.0]\n", STR(FN(pow_compress))); return 1; }
    if (preg<(T)0) { fprintf(stderr,"error in %s: preg must be nonnegative\n", STR(FN(pow_compress))); return 1; }
    for (size_t n=0u; n<N; ++n)
    {
        const T v = X[n] + preg;
        if (p==(T)1) { Y[n] = v; }
        else if (p==(T)0) { Y[n] = LOG(v); }
        else if (p==(T)0.5) { Y[n] = SQRT(v); }
        else if (FABS(p-(T)1/(T)3)<=(T)FLT_EPSILON) { Y[n] = CBRT(v); }
        else { Y[n] = POW(v,p); }
    }
    return 0;
}

/* ---- get_stft_freqs_s / _d : reference c/get_stft_freqs.c:17-31, :34-48 ---- */
int FN(get_stft_freqs)(T *Y, const size_t nfft, const size_t F, const T fs)
{
    if (F>nfft) { fprintf(stderr,"error in %s: F must be <= nfft\n", STR(FN(get_stft_freqs))); return 1; }
    const size_t Nyq = nfft/2u + 1u;
    const T finc = fs/(T)nfft;
    if (F>0u) { Y[0] = (T)0; }
    for (size_t f=1u; f<F && f<Nyq; ++f) { Y[f] = (T)f * finc; }
    for (size_t f=(Nyq>1u)?Nyq:1u; f<F; ++f) { Y[f] = -finc * (T)(nfft-f); }
    return 0;
}

#undef CAT_
#undef CAT
#undef FN
#undef FNC
#undef STR_
#undef STR
